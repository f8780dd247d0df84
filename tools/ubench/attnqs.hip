// attnqs.hip -- the transformer denoiser's attention kernels alone, on the library's own source (csrc/declayer.hip.h): the query-split
// k_attn_qs (+ k_attn_combine), one head geometry (8 heads x 64), T frames, batch B, every key split.  Prints us per launch (events around
// back-to-back launches), the max |error| against an fp64 CPU evaluation of sampled queries, and -- built with -DFDX_ATTN_TRACE[=2] -- where a
// wave's cycles go (s_memtime stamps).  (Round 4's key-split kernel ran in this harness until it was removed from the library: its numbers
// are in profiles/r05_attention_ubench_*.txt -- 29.3 us at T = 861, B = 1; 187 us at B = 8.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFDX_ATTN_TRACE -I fish_diffusion_amd/csrc -I include \
//         tools/ubench/attnqs.hip -o tools/ubench/attnqs && tools/ubench/attnqs [T=861] [B=1]
#include "declayer.hip.h"

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

using namespace fdx;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.f - 1.f; }

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 861, B = argc > 2 ? atoi(argv[2]) : 1;
  const int D = 512, DH = 64, ld = padded_ld(T, 64);
  const size_t n3 = (size_t)B * 3 * D * ld, n1 = (size_t)B * D * ld;
  std::vector<float> hq(n3, 0.f);
  unsigned seed = 12345;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < 3 * D; ++c)
      for (int t = 0; t < T; ++t) hq[((size_t)b * 3 * D + c) * ld + kHalo + t] = frand(seed) * (c < 2 * D ? 2.0f : 1.0f);
  std::vector<uint8_t> hmask((size_t)B * T, 0);
  for (int b = 0; b < B; ++b)
    for (int t = T - 37 - 11 * b; t < T; ++t) hmask[(size_t)b * T + t] = 1;     // a padded tail
  float *dq, *dO0, *dO1, *dP, *dML;
  uint8_t* dmask;
  attn_ksplit_forced() = 8;
  CHECK(hipMalloc(&dq, n3 * 4)); CHECK(hipMalloc(&dO0, n1 * 4)); CHECK(hipMalloc(&dO1, n1 * 4));
  CHECK(hipMalloc(&dP, (size_t)8 * B * D * ld * 4)); CHECK(hipMalloc(&dML, (size_t)8 * B * kHeads * 2 * round_up(T, 128) * 4));
  CHECK(hipMalloc(&dmask, hmask.size()));
  CHECK(hipMemcpy(dq, hq.data(), n3 * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dmask, hmask.data(), hmask.size(), hipMemcpyHostToDevice));
  CHECK(hipMemset(dO0, 0, n1 * 4)); CHECK(hipMemset(dO1, 0, n1 * 4));
  unsigned long long* dtrace = nullptr;
  const int max_wg = 8 * ((T + 127) / 128) * 8 * B;
  CHECK(hipMalloc(&dtrace, (size_t)max_wg * 4 * 8 * 8));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));

  auto args = [&](float* O, bool masked) {
    AttnArgs a{};
    const long bsD = (long)D * ld;
    a.Q = dq + kHalo; a.q_bs = 3 * bsD; a.ldq = ld;
    a.K = dq + kHalo + (size_t)D * ld; a.k_bs = 3 * bsD; a.ldk = ld;
    a.V = dq + kHalo + (size_t)2 * D * ld; a.v_bs = 3 * bsD; a.ldv = ld;
    a.O = O + kHalo; a.o_bs = bsD; a.ldo = ld; a.kmask = masked ? dmask : nullptr; a.Tq = T; a.Tk = T; a.scale = 1.f / sqrtf((float)DH);
    a.P = dP + kHalo; a.ML = dML;
#ifdef FDX_ATTN_TRACE
    a.trace = nullptr;
#endif
    return a;
  };
  auto time_it = [&](auto&& launch, int reps) {
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
  };
  const double flops = 4.0 * T * T * (double)D * B;
  printf("attention, 8 heads x 64, T = %d, B = %d: %.3f GFLOP per launch (%.2f us at the 157.3 TFLOP/s fp32 roof)\n", T, B, flops / 1e9, flops / 157.3e6);
  for (int masked = 0; masked < 2; ++masked) {
    printf("%s\n", masked ? "masked" : "unmasked");
    for (int ks : {1, 2, 3, 4, 5, 6, 8}) {
      if (ks > (T + 31) / 32) continue;
      attn_ksplit_forced() = ks;
      const AttnArgs a1 = args(dO1, masked);
      CHECK(hipMemset(dO1, 0, n1 * 4));
      const double t_new = time_it([&] { launch_attn_qs(DH, a1, B, s, nullptr); }, 200);
      // the attention kernel alone (no combine): launch by hand
      AttnArgs ak = a1;
      ak.B = B; ak.ksplit = ks; ak.TqR = round_up(T, 128); ak.p_split = (long)B * ak.o_bs;
      const dim3 grid(kHeads * ((T + 127) / 128) * ks, 1, B);
      const double t_k = time_it([&] { hipLaunchKernelGGL((k_attn_qs<64>), grid, dim3(256), 0, s, ak); }, 200);
      launch_attn_qs(DH, a1, B, s, nullptr);
      CHECK(hipStreamSynchronize(s));
      std::vector<float> o1(n1);
      CHECK(hipMemcpy(o1.data(), dO1, n1 * 4, hipMemcpyDeviceToHost));
      // fp64 reference on sampled (b, head, query)
      double emax = 0;
      for (int smp = 0; smp < 12; ++smp) {
        const int b = smp % B, h = (smp * 3) % 8, q = (smp * 977 + 5) % T;
        std::vector<double> sc(T);
        double mxs = -1e300;
        for (int k = 0; k < T; ++k) {
          double acc = 0;
          for (int d = 0; d < DH; ++d)
            acc += (double)hq[((size_t)b * 3 * D + h * DH + d) * ld + kHalo + q] * hq[((size_t)b * 3 * D + D + h * DH + d) * ld + kHalo + k];
          sc[k] = (masked && hmask[(size_t)b * T + k]) ? -1e300 : acc / 8.0;
          mxs = fmax(mxs, sc[k]);
        }
        double L = 0;
        for (int k = 0; k < T; ++k) { sc[k] = sc[k] < -1e299 ? 0.0 : exp(sc[k] - mxs); L += sc[k]; }
        for (int d = 0; d < DH; ++d) {
          double acc = 0;
          for (int k = 0; k < T; ++k) acc += sc[k] * hq[((size_t)b * 3 * D + 2 * D + h * DH + d) * ld + kHalo + k];
          emax = fmax(emax, fabs(acc / L - o1[((size_t)b * D + h * DH + d) * ld + kHalo + q]));
        }
      }
      printf("          k_attn_qs<64> keys split %d ways (%4d workgroups)%s  %7.2f us  %5.1f TFLOP/s = %4.1f %%   kernel alone %6.2f us = %4.1f %%   "
             "|kernel - fp64| %.2e\n", ks, (int)(grid.x * B), ks > 1 ? " + combine" : "          ", t_new, flops / t_new / 1e6,
             flops / t_new / 1e6 / 157.3 * 100, t_k, flops / t_k / 1e6 / 157.3 * 100, emax);
    }
  }
#ifdef FDX_ATTN_TRACE
  for (int ks : {1, 4}) {
    attn_ksplit_forced() = ks;
    AttnArgs ak = args(dO1, false);
    ak.B = B; ak.ksplit = ks; ak.TqR = round_up(T, 128); ak.p_split = (long)B * ak.o_bs;
    ak.trace = dtrace;
    const dim3 grid(kHeads * ((T + 127) / 128) * ks, 1, B);
    const size_t nw = (size_t)grid.x * B * 4;
    CHECK(hipMemset(dtrace, 0, nw * 8 * 8));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_attn_qs<64>), grid, dim3(256), 0, s, ak);
    CHECK(hipStreamSynchronize(s));
    std::vector<unsigned long long> tr(nw * 8);
    CHECK(hipMemcpy(tr.data(), dtrace, nw * 8 * 8, hipMemcpyDeviceToHost));
#if FDX_ATTN_TRACE == 2
    const char* names[] = {"Q loads issued -> Q scaled (first fabric round trip)", "first K / V tile staged + barrier", "(tile 0)",
                           "tile 1: next tile's loads issued + score product issued", "tile 1: softmax (mask, max, exp2, sum)",
                           "tile 1: second product issued", "tile 1: next tile staged into LDS + barrier"};
    const int last = 7;
#else
    const char* names[] = {"Q loads issued -> Q scaled (first fabric round trip)", "first K / V tile staged + barrier", "tile 0", "tile 1", "tile 2",
                           "remaining tiles", ""};
    const int last = 6;
#endif
    double sums[7] = {0}, tot = 0;
    size_t cnt = 0;
    for (size_t w = 0; w < nw; ++w) {
      const unsigned long long* t = &tr[w * 8];
      if (!t[0] || !t[last]) continue;
      unsigned long long prev = t[0];
      for (int k = 1; k <= last; ++k) {
        const unsigned long long cur = t[k] ? t[k] : prev;
        sums[k - 1] += (double)(cur - prev);
        prev = cur;
      }
      tot += (double)(t[last] - t[0]);
      ++cnt;
    }
    printf("k_attn_qs<64>, keys split %d ways: mean shader cycles per wave over %zu waves (stamps are s_memtime), first -> last stamp %.0f:\n", ks, cnt, tot / cnt);
    for (int k = 0; k < last; ++k) printf("    %-62s %9.0f\n", names[k], sums[k] / cnt);
  }
#endif
  return 0;
}
