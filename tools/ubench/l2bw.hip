// l2bw.hip -- what does ONE CU get from its XCD's L2 per clock, by access path?  (round 3: the fp16-split 64 x 64 kernel's K loop
// delivers ~25 B/clk/CU by LDS-DMA whatever the stage count or the number of issuing waves; is that the path's ceiling, and does the
// plain global_load -> VGPR path add to it?)
//   mode 0: global_load_dwordx4 -> VGPR          mode 1: global_load_lds_dwordx4 -> LDS          mode 2: both, half the bytes each
// Every workgroup (one per CU, 4 or 8 waves) streams the SAME `region` bytes (L2-resident after the first pass) `reps` times.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l2bw.hip -o tools/ubench/l2bw && tools/ubench/l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64, 1) void k_stream(const uint4* __restrict__ src, size_t region_groups, int reps, unsigned* sink, size_t wg_stride_groups) {
  __shared__ uint4 lds[NW * 64 * DEPTH];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint4* base = src + (size_t)blockIdx.x * wg_stride_groups;     // wg_stride 0: every workgroup reads the same region
  unsigned acc = 0;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)lds + (unsigned)wave * 1024u;
  const size_t step = (size_t)NW * 64;                                   // 16-byte groups per workgroup-wide request
  for (int r = 0; r < reps; ++r) {
    for (size_t g = 0; g + step * DEPTH <= region_groups; g += step * DEPTH) {
      uint4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const uint4* p = base + g + d * step + tid;
        const bool dma = MODE == 1 || (MODE == 2 && (d & 1));
        if (dma) {
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(p), "s"(lds0 + (unsigned)d * (NW * 1024u)) : "memory");
          v[d] = uint4{0, 0, 0, 0};
        } else {
          v[d] = *p;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc + lds[tid].x;
}

template <int MODE, int NW, int DEPTH>
static void run(const uint4* src, size_t region_bytes, unsigned* sink, int wgs, bool shared, double clk_ghz) {
  const size_t groups = region_bytes / 16;
  const int reps = 40;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int it = 0; it < 2; ++it) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_stream<MODE, NW, DEPTH>), dim3(wgs), dim3(NW * 64), 0, 0, src, groups, reps, sink, shared ? 0 : groups);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const size_t per_pass = groups / ((size_t)NW * 64 * DEPTH) * ((size_t)NW * 64 * DEPTH) * 16;
  const double bytes_per_wg = (double)per_pass * reps;
  const double gbs_cu = bytes_per_wg / (ms * 1e-3) / 1e9;
  printf("mode %d (%s)  waves %d  depth %2d  region %6zu KB %s  wgs %3d: %7.1f GB/s per CU = %5.1f B/clk at %.1f GHz;  chip %6.2f TB/s\n", MODE,
         MODE == 0 ? "load->VGPR" : MODE == 1 ? "LDS-DMA   " : "half/half ", NW, DEPTH, region_bytes >> 10, shared ? "shared " : "private", wgs, gbs_cu,
         gbs_cu / clk_ghz, clk_ghz, gbs_cu * wgs / 1e3);
}

int main() {
  const size_t total = (size_t)512 << 20;
  uint4* src; unsigned* sink;
  CHECK(hipMalloc(&src, total)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(src, 1, total));
  const double clk = 2.1;
  // one region shared by every workgroup (the weight slab of a row tile: L2 hits after the first toucher)
  for (size_t kb : {384, 1536}) {
    run<0, 4, 8>(src, kb << 10, sink, 256, true, clk);
    run<1, 4, 8>(src, kb << 10, sink, 256, true, clk);
    run<2, 4, 8>(src, kb << 10, sink, 256, true, clk);
    run<0, 4, 16>(src, kb << 10, sink, 256, true, clk);
    run<1, 4, 16>(src, kb << 10, sink, 256, true, clk);
    run<2, 4, 16>(src, kb << 10, sink, 256, true, clk);
    run<0, 8, 8>(src, kb << 10, sink, 256, true, clk);
    run<1, 8, 8>(src, kb << 10, sink, 256, true, clk);
    run<2, 8, 8>(src, kb << 10, sink, 256, true, clk);
  }
  // a private region per workgroup (streams from HBM / MALL: 256 x 1.5 MB = 384 MB)
  run<0, 4, 8>(src, (size_t)1536 << 10, sink, 256, false, clk);
  run<1, 4, 8>(src, (size_t)1536 << 10, sink, 256, false, clk);
  // fewer workgroups: is the ceiling per CU or per XCD / chip?
  run<1, 4, 8>(src, (size_t)384 << 10, sink, 64, true, clk);
  run<1, 4, 8>(src, (size_t)384 << 10, sink, 8, true, clk);
  run<0, 4, 8>(src, (size_t)384 << 10, sink, 8, true, clk);
  return 0;
}
