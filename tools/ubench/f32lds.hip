// f32lds.hip -- is an fp32 GEMM whose operands reach the MFMAs through LDS-DMA (global_load_lds_dwordx4 -> LDS -> ds_read) faster than the
// library's register-direct split-K tiles on the small-column problems of the widening rows?  (round 4, NOTES.md items 11-12: ConvNext's
// pwconv2 = 512 x 2048 x 861 runs 19.6 us, exactly what 786 KB of operands per workgroup cost at the 18 B/clk/CU of global_load -> VGPR.)
// Standalone: no library code.  C[M][N] = W[M][K] X[K][N], fp32, v_mfma_f32_32x32x2_f32.
//   workgroup tile 32 rows x 64 columns, 4 waves = 2 K-halves x 2 column blocks; K walked in super-stages of two 32-deep chunks (one per
//   K-half): A chunk 4 KB in fragment order [ks][half][row], B chunk 8 KB = 32 activation rows x 64 columns as they lie in memory;
//   NST super-stages of 24 KB in LDS, 6 one-KiB DMA pieces per wave per super-stage, own-piece vmcnt wait + one barrier per super-stage;
//   the two K-halves are summed through LDS at the end.
//   NKQ = 4 (8 waves, two per SIMD: 4 K-quarters x 2 column blocks, super-stages of four chunks, 144 KB of LDS, one workgroup per CU) is the
//   "second cut" the first measurement asked for; it was written AFTER the round's GPU budget ran out: compiled, never run.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f32lds.hip -o tools/ubench/f32lds && tools/ubench/f32lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int CHUNK_F = 1024 + 2048;   // floats per chunk: A 32 x 32 (fragment order) + B 32 rows x 64 columns
constexpr int PIECES = 6;              // 1-KiB DMA pieces per wave per super-stage: the two waves of a K-split load that split's chunk (12 KB)

struct Args {
  const float* Wp;   // packed: [mt][chunk][ks(16)][half(2)][row(32)]
  const float* X;    // [K][ld]
  float* C;          // [M][ld]
  int ld, n_chunks, n_tiles_n, n_mtiles, N;
};

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int NKQ, int NST>            // K-splits per workgroup (2 waves each): 2 = 4 waves, 4 = 8 waves; super-stages in LDS
__global__ __launch_bounds__(128 * NKQ, (NKQ * NST <= 6) ? 2 : 1) void k_f32lds(Args a) {
  constexpr int STAGE_F = NKQ * CHUNK_F;
  __shared__ float lds[NST * STAGE_F];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave >> 1, cb = wave & 1;            // K-split, column block
  const int half = lane >> 5, n = lane & 31;
  // tile -> XCD map: an XCD owns a contiguous run of tiles in row-major order (row runs)
  const int G = a.n_tiles_n * a.n_mtiles, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
  const int q8 = G >> 3, r8 = G & 7;
  const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int mt = L / a.n_tiles_n, nt = L - mt * a.n_tiles_n;
  const int t0 = nt * 64;
  const int n_ss = (a.n_chunks + NKQ - 1) / NKQ;       // super-stages

  const float* Ag = a.Wp + (size_t)mt * a.n_chunks * 1024;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)lds;
  auto glds16 = [&](const float* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  // super-stage layout (floats): [chunk h][A 1024 | B 2048].  Pieces of a chunk: A 4 (1 KiB each, linear), B 8 (4 rows x 64 columns each).
  // wave (kh, cb) issues, for chunk h = kh of the super-stage: A pieces 2 cb, 2 cb + 1 and B pieces 4 cb .. 4 cb + 3.
  auto issue = [&](int ss, int st) {
    ss = min(ss, n_ss - 1);                            // the tail re-loads the last super-stage (uniform vmcnt accounting); harmless
    const int c = min(NKQ * ss + kh, a.n_chunks - 1);
    const unsigned base = lds0 + (unsigned)(st * STAGE_F + kh * CHUNK_F) * 4u;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int piece = 2 * cb + p;
      glds16(Ag + (size_t)c * 1024 + piece * 256 + lane * 4, base + piece * 1024u);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int piece = 4 * cb + p;                    // rows 4 piece .. 4 piece + 3 of the chunk
      const int row = piece * 4 + (lane >> 4), col = (lane & 15) * 4;
      glds16(a.X + (size_t)(c * 32 + row) * a.ld + t0 + col, base + 4096u + piece * 1024u);
    }
  };

  f16v acc, acc1;                                      // two accumulator chains (even / odd k-steps), summed at the end
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s, s);
  for (int j = 0; j < n_ss; ++j) {
    static_assert((NST - 2) * PIECES <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PIECES) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(j + NST - 1, (j + NST - 1) % NST);
    if (NKQ * j + kh < a.n_chunks) {
      const float* la = lds + (j % NST) * STAGE_F + kh * CHUNK_F + half * 32 + n;
      const float* lb = la - (half * 32 + n) + 1024 + half * 64 + cb * 32 + n;
      float av[16], bv[16];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) { av[ks] = la[ks * 64]; bv[ks] = lb[ks * 128]; }
#ifndef F32LDS_NO_SCHED_BARRIER
      // all 16 fragment reads are in flight before the first MFMA: left alone, hipcc sinks each B read next to its use and waits lgkmcnt(0) in
      // front of every MFMA pair (one exposed LDS latency per 128 MFMA cycles -- what the first three cuts measured)
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int ks = 0; ks < 16; ks += 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], bv[ks], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks + 1], bv[ks + 1], acc1, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing re-loads landed before the LDS is reused / released
  __syncthreads();
  // ---- sum the K-splits through LDS, store
  if (kh != 0) {
    float* red = lds + ((kh - 1) * 2 + cb) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (kh == 0) {
    const int col = t0 + cb * 32 + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
#pragma unroll
      for (int q = 1; q < NKQ; ++q) v += lds[((q - 1) * 2 + cb) * 1024 + r * 64 + lane];
      if (col < a.N) a.C[(size_t)(mt * 32 + acc_row(r, half)) * a.ld + col] = v;
    }
  }
}

template <int NKQ, int NST>
static void run(int M, int K, int N, const char* what, double ref_us) {
  auto kern = k_f32lds<NKQ, NST>;
  const int n_tiles_n = (N + 63) / 64, ld = n_tiles_n * 64 + 64, n_chunks = K / 32, n_mtiles = M / 32;
  std::vector<float> W((size_t)M * K), X((size_t)K * ld, 0.f), Wp((size_t)M * K);
  unsigned s = 4321;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : W) v = rnd();
  for (int k = 0; k < K; ++k)
    for (int t = 0; t < N; ++t) X[(size_t)k * ld + t] = rnd();
  for (int mt = 0; mt < n_mtiles; ++mt)
    for (int c = 0; c < n_chunks; ++c)
      for (int ks = 0; ks < 16; ++ks)
        for (int h = 0; h < 2; ++h)
          for (int row = 0; row < 32; ++row)
            Wp[(((size_t)mt * n_chunks + c) * 32 + ks * 2 + h) * 32 + row] = W[(size_t)(mt * 32 + row) * K + c * 32 + 2 * ks + h];
  float *dW, *dX, *dC;
  CHECK(hipMalloc(&dW, Wp.size() * 4)); CHECK(hipMalloc(&dX, X.size() * 4 + 4096)); CHECK(hipMalloc(&dC, (size_t)M * ld * 4));
  CHECK(hipMemcpy(dW, Wp.data(), Wp.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(dC, 0, (size_t)M * ld * 4));
  Args a{dW, dX, dC, ld, n_chunks, n_tiles_n, n_mtiles, N};
  const int grid = n_tiles_n * n_mtiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * NKQ), 0, 0, a);
  CHECK(hipDeviceSynchronize());
  std::vector<float> C((size_t)M * ld);
  CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int i = 0; i < 4000; ++i) {
    const int m = (int)((unsigned)(i * 2654435761u) % (unsigned)M), t = (int)((unsigned)(i * 40503u + 17) % (unsigned)N);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)W[(size_t)m * K + k] * (double)X[(size_t)k * ld + t];
    worst = fmax(worst, fabs(ref - (double)C[(size_t)m * ld + t]));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int reps = 400;
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * NKQ), 0, 0, a);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * NKQ), 0, 0, a);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, tf = 2.0 * M * K * N / (us * 1e-6) / 1e12;
  printf("NKQ %d NST %d  %-44s M %4d K %4d N %4d  %4d workgroups  %7.2f us per launch (back to back)  %6.1f TFLOP/s (%4.1f %% of 157.3)  max abs err %.2e  | library kernel: %.1f us\n",
         NKQ, NST, what, M, K, N, grid, us, tf, tf / 157.3 * 100, worst, ref_us);
  CHECK(hipFree(dW)); CHECK(hipFree(dX)); CHECK(hipFree(dC));
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  if (mode == 6) {                       // third cut: six super-stages in flight (144 KB, one workgroup per CU)
    run<2, 6>(512, 2048, 861, "ConvNext pwconv2 / transformer linear2", 19.6);
    run<2, 6>(1024, 512, 861, "WaveNet out-projection (GEMM part)", 11.8);
    run<2, 6>(2048, 512, 861, "ConvNext pwconv1 / transformer linear1", 26.2);
    run<2, 6>(512, 2048, 8 * 861, "pwconv2 at batch 8 (one row of 6888 cols)", 0.0);
    return 0;
  }
  run<2, 3>(512, 2048, 861, "ConvNext pwconv2 / transformer linear2", 19.6);
  run<2, 3>(2048, 512, 861, "ConvNext pwconv1 / transformer linear1", 26.2);
  run<2, 3>(1024, 512, 861, "WaveNet out-projection (GEMM part)", 11.8);
  run<2, 3>(512, 512, 861, "attention out-proj / WaveNet skip proj", 6.5);
  run<2, 3>(512, 2048, 8 * 861, "pwconv2 at batch 8 (one row of 6888 cols)", 0.0);
  if (mode == 8) {                       // second cut: 8 waves
    run<4, 3>(512, 2048, 861, "ConvNext pwconv2 / transformer linear2", 19.6);
    run<4, 3>(2048, 512, 861, "ConvNext pwconv1 / transformer linear1", 26.2);
    run<4, 3>(1024, 512, 861, "WaveNet out-projection (GEMM part)", 11.8);
    run<4, 3>(512, 2048, 8 * 861, "pwconv2 at batch 8 (one row of 6888 cols)", 0.0);
  }
  return 0;
}
