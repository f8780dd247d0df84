// clkwait.hip -- round 5, VERDICT r4 item 4(a).  The library's instrumented residual-block kernels read 2.13 / 2.19 GHz on s_memtime / s_memrealtime
// where an MFMA-only loop reads 2.39 on the same two counters (profiles/r04_mfma_clock_ubench.txt, r04_ktrace_headline_fp32.txt).  Is that a lower
// CLOCK (power management under the real kernels' mix) or does the shader-clock counter simply not advance in some wait states?  One decisive
// variant per suspect, each as ONE long launch (~0.2 s) and as a chain of ~25 us launches, every wave stamping both counters at its start and end:
//   0  v_mfma_f32_16x16x4_f32 only                                    (baseline: 2.39 GHz expected)
//   1  s_sleep only                                                    (a wave that does nothing but wait: does s_memtime advance?)
//   2  MFMAs with a workgroup barrier every 16                         (barrier wait states)
//   3  MFMA bursts alternating with dependent cold loads (256 MB)      (s_waitcnt vmcnt wait states: ~half the lifetime spent waiting)
//   4  MFMAs + the kernels' LDS reduction (ds_write, barrier, ds_read) every 64 MFMAs
//   5  MFMAs + an exp / rcp-heavy VALU phase every 64 MFMAs            (the gate epilogue's mix)
//   6  2 + 3 + 4 + 5 together
//   7  MFMAs fed from registers while the wave also streams 16 x 1 KiB from an L2-resident 2 MB region and 16 x 1 KiB out of LDS per 64 MFMAs
//      (the real K loops' simultaneous load on matrix pipes, L2 fabric and LDS: does the reading follow board POWER?); 8 = 7 with two workgroups per CU
// Reported per variant: d(s_memtime) / d(s_memrealtime) x 100 MHz per wave (mean, min, max).  Board power / sclk are sampled from sysfs by the
// calling script (tools/r05_run6.sh) against the wall-clock window printed here.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clkwait.hip -o tools/ubench/clkwait && tools/ubench/clkwait
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long c0, c1, r0, r1; };

template <int V>
__global__ __launch_bounds__(256, 2) void k_var(const unsigned* __restrict__ chase, unsigned chase_mask, const float* __restrict__ src, int iters,
                                                float* sink, Stamp* stamps) {
  __shared__ float red[4 * 64 * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_global = blockIdx.x * 4 + wave;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a = src[tid], b = src[tid + 256];
  f4 acc[8] = {};
  unsigned p = (wave_global * 7919u + lane * 64u) & chase_mask;
  float vsum = 0.f;
  if constexpr (V == 7) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = a * (float)(j + 1);
    __syncthreads();
  }
  for (int i = 0; i < iters; ++i) {
    if constexpr (V == 1) {
      __builtin_amdgcn_s_sleep(127);
      __builtin_amdgcn_s_sleep(127);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(j & 1 ? a : b, j & 2 ? a : b, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(j & 1 ? b : a, j & 2 ? b : a, acc[j], 0, 0, 0);
        if constexpr (V == 2 || V == 6) __syncthreads();
      }
      if constexpr (V == 7) {   // independent 16-byte loads (L2 hits) + LDS reads, consumed with a vanishing weight so nothing is dropped
        const f4* g4 = reinterpret_cast<const f4*>(chase);
        f4 gs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const f4 gv = g4[((size_t)(i * 16 + k) * 4099u + (size_t)wave_global * 64u + lane) & (size_t)0x1FFFF];   // 2 MB region
          const f4 lv = *reinterpret_cast<const f4*>(&red[((k * 64 + lane) * 4) & 2047]);
          gs += gv + lv;
        }
        a += (gs[0] + gs[1] + gs[2] + gs[3]) * 1e-38f;
      }
      if constexpr (V == 3 || V == 6) {   // 3 dependent loads from a 256 MB region: ~3 HBM round trips of pure waiting
#pragma unroll
        for (int k = 0; k < 3; ++k) p = chase[p] & chase_mask;
        a += (float)(p & 1) * 1e-30f;
      }
      if constexpr (V == 4 || V == 6) {   // fixed-order cross-wave reduction as in convgemm16s: write, barrier, read the four partials
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(wave * 64 + lane) * 8 + j] = acc[j][0];
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int j = 0; j < 8; ++j) s += red[(w * 64 + lane) * 8 + j];
        vsum += s;
        __syncthreads();
      }
      if constexpr (V == 5 || V == 6) {   // sigmoid * tanh on 16 values: v_exp_f32 / v_rcp_f32 heavy
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float g = acc[j][1], f = acc[j][2];
          vsum += __builtin_amdgcn_rcpf(1.f + __expf(-g)) * (1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * f) + 1.f));
        }
      }
    }
  }
  float s = vsum + (float)p;
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 1.2345f) sink[0] = s;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) stamps[wave_global] = Stamp{c0, c1, r0, r1};
}

static double now_s() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }

template <int V>
static void run(const unsigned* chase, unsigned mask, const float* src, float* sink, Stamp* stamps_d, int iters_long, int iters_short, const char* what, int wgs = 256) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  auto report = [&](const char* regime, int n_launch, float ms, double t0, double t1) {
    std::vector<Stamp> st((size_t)n_launch * wgs * 4);
    CHECK(hipMemcpy(st.data(), stamps_d, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    double g = 0, lo = 1e9, hi = 0, life = 0;
    for (auto& q : st) {
      const double v = double(q.c1 - q.c0) / double(q.r1 - q.r0) * 0.1;
      g += v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; life += double(q.r1 - q.r0) * 10.0;
    }
    printf("variant %d %-58s %-6s %4d launch(es) %9.3f ms  wave lifetime %9.2f us  memtime/memrealtime = %.3f GHz (min %.3f max %.3f)  wall %.3f .. %.3f\n", V, what,
           regime, n_launch, ms, life / st.size() * 1e-3, g / st.size(), lo, hi, t0, t1);
  };
  float ms = 0;
  for (int it = 0; it < 2; ++it) {   // long
    const double t0 = now_s();
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_var<V>), dim3(wgs), dim3(256), 0, 0, chase, mask, src, iters_long, sink, stamps_d);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
    if (it == 1) report("long", 1, ms, t0, now_s());
  }
  const int n = 400;
  for (int it = 0; it < 2; ++it) {   // chain of short launches
    const double t0 = now_s();
    CHECK(hipEventRecord(a));
    for (int l = 0; l < n; ++l) hipLaunchKernelGGL((k_var<V>), dim3(wgs), dim3(256), 0, 0, chase, mask, src, iters_short, sink, stamps_d + (size_t)l * wgs * 4);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
    if (it == 1) report("chain", n, ms, t0, now_s());
  }
  CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
}

int main() {
  const size_t n_chase = (size_t)64 << 20;   // 256 MB of indices
  unsigned* chase; float* src; float* sink; Stamp* stamps;
  CHECK(hipMalloc(&chase, n_chase * 4)); CHECK(hipMalloc(&src, 4096)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&stamps, sizeof(Stamp) * 512 * 4 * 400));
  std::vector<unsigned> h(n_chase);
  unsigned s = 777;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 4) & (unsigned)(n_chase - 1); }
  CHECK(hipMemcpy(chase, h.data(), n_chase * 4, hipMemcpyHostToDevice));
  std::vector<float> hs(1024);
  for (auto& v : hs) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  CHECK(hipMemcpy(src, hs.data(), 4096, hipMemcpyHostToDevice));
  const unsigned mask = (unsigned)(n_chase - 1);
  // iteration counts: 64 MFMAs of 32 cycles per iteration = 2048 cycles = ~0.85 us at 2.4 GHz (+ the variant's extra phase)
  run<0>(chase, mask, src, sink, stamps, 240000, 29, "MFMA only");
  run<1>(chase, mask, src, sink, stamps, 30000, 4, "s_sleep only (no work)");
  run<2>(chase, mask, src, sink, stamps, 240000, 29, "MFMA + s_barrier every 16");
  run<3>(chase, mask, src, sink, stamps, 60000, 8, "MFMA bursts + 3 dependent cold loads per 64 MFMAs");
  run<4>(chase, mask, src, sink, stamps, 200000, 24, "MFMA + LDS reduction (write, barrier, read) per 64");
  run<5>(chase, mask, src, sink, stamps, 200000, 24, "MFMA + exp / rcp VALU phase per 64");
  run<6>(chase, mask, src, sink, stamps, 50000, 6, "barriers + cold loads + LDS reduction + exp phase");
  run<7>(chase, mask, src, sink, stamps, 200000, 24, "MFMA + 16 KiB from L2 + 16 KiB from LDS per 64 MFMAs");
  run<7>(chase, mask, src, sink, stamps, 100000, 12, "same, two workgroups per CU", 512);
  run<0>(chase, mask, src, sink, stamps, 120000, 15, "MFMA only, two workgroups per CU", 512);
  return 0;
}
