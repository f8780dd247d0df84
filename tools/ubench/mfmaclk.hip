// mfmaclk.hip -- what do the fp32 matrix pipes of THIS chip sustain, and at what shader clock, by instruction mix?
// (round 4, VERDICT r3 item 5: the "2.05 GHz / 134 TFLOP/s attainable" reading of the residual-block kernels came from one
// instrumented forward; the guide measures 155 TFLOP/s on the same instructions.  Independent evidence, no library code.)
//
//   mode 0: v_mfma_f32_16x16x4_f32 only        (NACC independent accumulators per wave, random register operands)
//   mode 1: the residual-block K loop's mix     6 global_load_dwordx4 per 16 MFMAs, operands from the loads, L2-resident region
//   mode 2: mode 1 + 3 of the 6 loads re-read the same line (the taps' L1 re-reads)
//   mode 3: v_mfma_f32_32x32x2_f32 only
// Every mode runs WAVES waves per SIMD on all 256 CUs for >= 10 ms; each wave stamps s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz) at start and end: effective clock = d(memtime) / d(realtime) * 100 MHz, per wave, averaged.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfmaclk.hip -o tools/ubench/mfmaclk && tools/ubench/mfmaclk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long c0, c1, r0, r1; };

__device__ __forceinline__ unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ unsigned long long ct() { return __builtin_amdgcn_s_memtime(); }

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_mfma(const f4* __restrict__ src, size_t region_groups, int iters, float* sink, Stamp* stamps) {
  const int tid = threadIdx.x;
  const int wave_global = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
  const unsigned long long c0 = ct(), r0 = rt();
  float a = src[tid].x, b = src[tid + 64].y;
  if constexpr (MODE == 3) {
    f16v acc[2] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc[0][j] + acc[1][j];
    if (s == 1.2345f) sink[0] = s;
  } else {
    f4 acc[8] = {};
    if constexpr (MODE == 0) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(j & 1 ? a : b, j & 2 ? a : b, acc[j], 0, 0, 0);
        }
      }
    } else {
      // 6 loads per 16 MFMAs: 2 "weight" vectors (4 k-steps x the lane's row) + 4 "activation" vectors; a 3-deep register ring
      const size_t mask = region_groups - 1;                 // power of two
      size_t cur = ((size_t)wave_global * 6151u + tid) & mask;
      f4 ring[3][6];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int l = 0; l < 6; ++l) {
          const size_t off = (MODE == 2 && l >= 3) ? (cur + (size_t)(l - 3) * 64) : (cur + (size_t)l * 64);
          ring[s][l] = src[off & mask];
          if (l == 5) cur = (cur + 6 * 64 + 4096 * 3) & mask;
        }
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int ld = (s + 2) % 3;
#pragma unroll
          for (int l = 0; l < 6; ++l) {
            const size_t off = (MODE == 2 && l >= 3) ? (cur + (size_t)(l - 3) * 64) : (cur + (size_t)l * 64);
            ring[ld][l] = src[off & mask];
          }
          cur = (cur + 6 * 64 + 4096 * 3) & mask;
          __builtin_amdgcn_sched_barrier(0);     // the loads of stage s+2 are issued BEFORE this stage's 16 MFMAs (hipcc sinks them otherwise)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const float wa = ring[s][kk >> 1][(kk & 1) * 2], wb = ring[s][kk >> 1][(kk & 1) * 2 + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j + (kk & 1) * 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(j & 1 ? wa : wb, ring[s][2 + kk][j], acc[j + (kk & 1) * 4], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 1.2345f) sink[0] = s;
  }
  const unsigned long long c1 = ct(), r1 = rt();
  if ((tid & 63) == 0) stamps[wave_global] = Stamp{c0, c1, r0, r1};
}

template <int MODE>
static void run(const f4* src, size_t region_bytes, float* sink, Stamp* stamps_d, int waves_per_simd, int iters, const char* what) {
  const int cus = 256, wgs = cus * waves_per_simd;     // 4 waves per workgroup = one per SIMD; `waves_per_simd` workgroups per CU
  size_t groups = 1;
  while (groups * 2 * 16 <= region_bytes) groups *= 2;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_mfma<MODE>), dim3(wgs), dim3(256), 0, 0, src, groups, iters, sink, stamps_d);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  std::vector<Stamp> st(wgs * 4);
  CHECK(hipMemcpy(st.data(), stamps_d, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
  double ghz = 0, cyc = 0;
  for (auto& s : st) { ghz += double(s.c1 - s.c0) / double(s.r1 - s.r0) * 0.1; cyc += double(s.c1 - s.c0); }
  ghz /= st.size(); cyc /= st.size();
  // flops: 16x16x4 = 2*16*16*4 = 2048 per MFMA per wave; 32x32x2 = 4096
  double mfmas_per_wave = MODE == 3 ? 16.0 * iters : (MODE == 0 ? 16.0 * iters : 48.0 * iters);
  double fl = mfmas_per_wave * (MODE == 3 ? 4096.0 : 2048.0) * wgs * 4;
  double cyc_per_mfma = cyc / mfmas_per_wave * 1.0;
  printf("mode %d %-44s waves/SIMD %d  %8.3f ms  %7.2f TFLOP/s (%5.1f %% of 157.3)  clock %.3f GHz  -> %.1f TFLOP/s at that clock;  %.1f shader cycles per MFMA per wave\n",
         MODE, what, waves_per_simd, ms, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 1e12 / 157.3 * 100, ghz, ghz * 64 * 1024 / 1e3, cyc_per_mfma);
  CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
}

// Short launches in a chain (what the sampler's hipGraph is: 25 us kernels back to back): N launches on one stream, no host sync between
// them; the effective clock of the waves of EVERY launch (d memtime / d memrealtime per wave, one stamp slot per launch) and the launch period.
template <int MODE>
static void run_chain(const f4* src, size_t region_bytes, float* sink, Stamp* stamps_d, int iters, int n_launch, const char* what) {
  const int wgs = 256;
  size_t groups = 1;
  while (groups * 2 * 16 <= region_bytes) groups *= 2;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    CHECK(hipEventRecord(a));
    for (int l = 0; l < n_launch; ++l)
      hipLaunchKernelGGL((k_mfma<MODE>), dim3(wgs), dim3(256), 0, 0, src, groups, iters, sink, stamps_d + (size_t)l * wgs * 4);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  std::vector<Stamp> st((size_t)n_launch * wgs * 4);
  CHECK(hipMemcpy(st.data(), stamps_d, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
  double ghz = 0, ns = 0, ghz_first = 0, ghz_last = 0;
  for (size_t i = 0; i < st.size(); ++i) {
    const double g = double(st[i].c1 - st[i].c0) / double(st[i].r1 - st[i].r0) * 0.1;
    ghz += g; ns += double(st[i].r1 - st[i].r0) * 10.0;
    if (i < (size_t)wgs * 4) ghz_first += g;
    if (i >= st.size() - (size_t)wgs * 4) ghz_last += g;
  }
  // span of one launch: first wave start -> last wave end (realtime), averaged over the launches; gap = period - span
  double span = 0;
  for (int l = 0; l < n_launch; ++l) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < wgs * 4; ++w) { const Stamp& q = st[(size_t)l * wgs * 4 + w]; lo = q.r0 < lo ? q.r0 : lo; hi = q.r1 > hi ? q.r1 : hi; }
    span += double(hi - lo) * 10.0;
  }
  printf("chain mode %d %-40s %4d launches  period %6.2f us  span %6.2f us  wave lifetime %6.2f us  clock %.3f GHz (first launch %.3f, last %.3f)\n", MODE, what,
         n_launch, ms * 1e3 / n_launch, span / n_launch * 1e-3, ns / st.size() * 1e-3, ghz / st.size(), ghz_first / (wgs * 4), ghz_last / (wgs * 4));
  CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
}

int main(int argc, char** argv) {
  const size_t total = (size_t)64 << 20;
  f4* src; float* sink; Stamp* stamps;
  CHECK(hipMalloc(&src, total)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&stamps, sizeof(Stamp) * 256 * 4 * 512));
  std::vector<float> h(total / 4);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }   // random data: DVFS sees realistic toggling
  CHECK(hipMemcpy(src, h.data(), total, hipMemcpyHostToDevice));
  const int scale = argc > 1 ? atoi(argv[1]) : 1;
  for (int w : {1, 2}) {
    run<0>(src, 1 << 20, sink, stamps, w, 200000 * scale / w, "v_mfma_f32_16x16x4_f32 only");
    run<3>(src, 1 << 20, sink, stamps, w, 100000 * scale / w, "v_mfma_f32_32x32x2_f32 only");
    run<1>(src, 2 << 20, sink, stamps, w, 60000 * scale / w, "16x16x4 + 6 dwordx4 loads / 16 MFMA (L2, 2 MB)");
    run<2>(src, 2 << 20, sink, stamps, w, 60000 * scale / w, "same, 3 of 6 loads re-read a line (L1 hits)");
    run<1>(src, 48 << 20, sink, stamps, w, 60000 * scale / w, "same as mode 1, 48 MB region (MALL / HBM)");
  }
  // short launches back to back (the regime of the sampler's kernels: ~12 / ~25 us per launch)
  if (argc > 2) {
    run_chain<0>(src, 1 << 20, sink, stamps, 117, 400, "16x16x4 only, ~25 us launches");
    run_chain<0>(src, 1 << 20, sink, stamps, 56, 400, "16x16x4 only, ~12 us launches");
    run_chain<1>(src, 2 << 20, sink, stamps, 29, 400, "K-loop load mix (L2), ~25 us launches");
    run_chain<1>(src, 48 << 20, sink, stamps, 29, 400, "K-loop load mix (48 MB), ~25 us launches");
    run_chain<0>(src, 1 << 20, sink, stamps, 4680, 100, "16x16x4 only, ~1 ms launches");
    return 0;
  }
  // the same MFMA-only loop on zero operands (the guide's DVFS note: zero data clocks higher)
  CHECK(hipMemset(src, 0, total));
  run<0>(src, 1 << 20, sink, stamps, 1, 200000 * scale, "16x16x4 only, ZERO operands");
  return 0;
}
