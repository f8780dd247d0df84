// rbfused.hip -- the fused small-channel ResBlock1 kernel (csrc/resblock_fused.hip.h) on its own: launch time per (C, k) at the vocoder's
// geometry and, per wave, the shader-clock time of each phase (window load, the six conv passes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFDX_RB_TRACE tools/ubench/rbfused.hip -o tools/ubench/rbfused && tools/ubench/rbfused
#include "../../fish_diffusion_amd/csrc/resblock_fused.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace fdx;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
thread_local std::string fdx::g_last_error;

static void run(int C, int KS, int L, int B, int d2mode) {
  const int ld = 32 + (L + 255) / 256 * 256 + 32;
  const size_t act = (size_t)B * C * ld;
  std::vector<float> hx(act), hw(6 * rb_fused_floats(C, KS)), hb(6 * C);
  unsigned s = 1234;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() * (1.5f / sqrtf((float)C * KS));
  for (auto& v : hb) v = rnd() * 0.1f;
  float *X, *O, *Wd, *Bd; unsigned long long* tr;
  CHECK(hipMalloc(&X, act * 4)); CHECK(hipMalloc(&O, act * 4)); CHECK(hipMalloc(&Wd, hw.size() * 4)); CHECK(hipMalloc(&Bd, hb.size() * 4));
  CHECK(hipMemcpy(X, hx.data(), act * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(O, 0, act * 4));
  CHECK(hipMemcpy(Wd, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(Bd, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  RbFusedArgs a{};
  a.X = X + 32; a.x_bs = (long)C * ld; a.ldx = ld; a.out = O + 32; a.o_bs = (long)C * ld; a.ldo = ld; a.W = Wd; a.bias = Bd; a.L = L;
  const int dil[3] = {1, 3, 5};
  for (int j = 0; j < 3; ++j) { a.d1[j] = dil[j]; a.d2[j] = d2mode ? dil[j] : 1; }
  a.slope = 0.1f; a.mode = 2; a.div = 3.f;
  const int H = rb_halo(KS, a.d1, a.d2), H4 = (H + 3) & ~3;
  const int N = rb_pick_n(C, H4, L, B);
  const int tiles = B * ((L + N - 1) / N);
  CHECK(hipMalloc(&tr, (size_t)tiles * kRbWaves * 8 * 8));
  a.trace = tr;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms = 0;
  for (int it = 0; it < 3; ++it) {
    CHECK(hipEventRecord(e0));
    CHECK(launch_resblock1_fused(C, KS, a, B, nullptr));
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<unsigned long long> st((size_t)tiles * kRbWaves * 8);
  CHECK(hipMemcpy(st.data(), tr, st.size() * 8, hipMemcpyDeviceToHost));
  double ph[7] = {0};
  for (size_t w = 0; w < (size_t)tiles * kRbWaves; ++w)
    for (int i = 0; i < 7; ++i) ph[i] += (double)(st[w * 8 + i + 1] - st[w * 8 + i]);
  for (double& v : ph) v /= tiles * (double)kRbWaves;
  double flops = 0; { int rem = H; const int KH = (KS - 1) / 2; for (int j = 0; j < 3; ++j) { flops += 2.0 * 2.0 * C * C * KS * (double)L * B; (void)rem; (void)KH; } }
  printf("C %2d k %2d L %7d B %2d d2 %s: N %4d (H %3d) tiles %4d  %8.1f us  %6.1f TFLOP/s algorithmic (%4.1f %%) | cycles per wave: load %6.0f c1_0 %6.0f c2_0 %6.0f c1_1 %6.0f c2_1 %6.0f c1_2 %6.0f c2_2 %6.0f\n",
         C, KS, L, B, d2mode ? "=d1" : "1", N, H, tiles, ms * 1e3, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3 * 100, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
  CHECK(hipFree(X)); CHECK(hipFree(O)); CHECK(hipFree(Wd)); CHECK(hipFree(Bd)); CHECK(hipFree(tr));
}

// CPU restatement of one ResBlock1 (models.py:103-110) + the output mode, double accumulation: the harness' own correctness check
static void check(int C, int KS, int L, int d2mode) {
  const int B = 1, ld = 32 + (L + 255) / 256 * 256 + 32;
  std::vector<float> hx((size_t)C * ld, 0.f), hw(6 * rb_fused_floats(C, KS)), hwr(6 * (size_t)C * C * KS), hb(6 * C), ho((size_t)C * ld, 0.f);
  unsigned s = 99;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (int c = 0; c < C; ++c) for (int t = 0; t < L; ++t) hx[(size_t)c * ld + 32 + t] = rnd();
  for (auto& v : hwr) v = rnd() * (1.5f / sqrtf((float)C * KS));
  for (auto& v : hb) v = rnd() * 0.1f;
  for (int i = 0; i < 6; ++i) rb_fused_pack(hw.data() + i * rb_fused_floats(C, KS), hwr.data() + (size_t)i * C * C * KS, C, KS);
  for (int c = 0; c < C; ++c) for (int t = 0; t < L; ++t) ho[(size_t)c * ld + 32 + t] = rnd();     // the running MRF sum (mode 2 reads it)
  const int dil[3] = {1, 3, 5};
  auto lrelu = [](double v) { return v > 0 ? v : v * (double)0.1f; };
  std::vector<double> x((size_t)C * L), t1((size_t)C * L), t2((size_t)C * L);
  for (int c = 0; c < C; ++c) for (int t = 0; t < L; ++t) x[(size_t)c * L + t] = hx[(size_t)c * ld + 32 + t];
  auto conv = [&](const std::vector<double>& in, std::vector<double>& out, int ci, int d) {
    for (int r = 0; r < C; ++r) for (int t = 0; t < L; ++t) {
      double a = hb[ci * C + r];
      for (int c = 0; c < C; ++c) for (int k = 0; k < KS; ++k) {
        const int u = t + (k - (KS - 1) / 2) * d;
        if (u >= 0 && u < L) a += (double)hwr[(((size_t)ci * C + r) * C + c) * KS + k] * lrelu(in[(size_t)c * L + u]);
      }
      out[(size_t)r * L + t] = a;
    }
  };
  for (int j = 0; j < 3; ++j) {
    conv(x, t1, 2 * j, dil[j]);
    conv(t1, t2, 2 * j + 1, d2mode ? dil[j] : 1);
    for (size_t i = 0; i < x.size(); ++i) x[i] = t2[i] + x[i];
  }
  float *X, *O, *Wd, *Bd;
  CHECK(hipMalloc(&X, hx.size() * 4)); CHECK(hipMalloc(&O, ho.size() * 4)); CHECK(hipMalloc(&Wd, hw.size() * 4)); CHECK(hipMalloc(&Bd, hb.size() * 4));
  CHECK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(O, ho.data(), ho.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(Wd, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(Bd, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  RbFusedArgs a{};
  a.X = X + 32; a.x_bs = (long)C * ld; a.ldx = ld; a.out = O + 32; a.o_bs = (long)C * ld; a.ldo = ld; a.W = Wd; a.bias = Bd; a.L = L;
  for (int j = 0; j < 3; ++j) { a.d1[j] = dil[j]; a.d2[j] = d2mode ? dil[j] : 1; }
  a.slope = 0.1f; a.mode = 2; a.div = 3.f; a.trace = nullptr;
  CHECK(launch_resblock1_fused(C, KS, a, B, nullptr));
  std::vector<float> got(ho.size());
  CHECK(hipMemcpy(got.data(), O, got.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, big = 0;
  for (int c = 0; c < C; ++c) for (int t = 0; t < L; ++t) {
    const double want = ((double)ho[(size_t)c * ld + 32 + t] + x[(size_t)c * L + t]) / 3.0;
    worst = fmax(worst, fabs(want - got[(size_t)c * ld + 32 + t])); big = fmax(big, fabs(want));
  }
  printf("check C %2d k %2d L %5d d2 %s: max abs err %.3e (max |value| %.2f)  %s\n", C, KS, L, d2mode ? "=d1" : "1", worst, big, worst < 1e-4 * fmax(1.0, big) ? "OK" : "MISMATCH");
  CHECK(hipFree(X)); CHECK(hipFree(O)); CHECK(hipFree(Wd)); CHECK(hipFree(Bd));
}

int main() {
  for (int C : {16, 32}) for (int ks : {3, 7, 11}) for (int L : {40, 1536, 3001}) check(C, ks, L, 0);
  check(32, 11, 2000, 1);
  for (int ks : {3, 7, 11}) run(16, ks, 440832, 1, 0);
  for (int ks : {3, 7, 11}) run(32, ks, 220416, 1, 0);
  run(16, 11, 440832, 8, 0);
  run(32, 11, 440832, 1, 1);    // RefineGAN's last up stage: both convs of a pair dilated
  return 0;
}
