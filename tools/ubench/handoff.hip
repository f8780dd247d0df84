// handoff.hip -- what does a producer -> consumer hand-off between two dependent phases cost on an MI355X, by mechanism?
// (round 3: the residual-block launches of the batch-1 sampler are ~5 us of fixed cost each, 4000 times per utterance; DESIGN.md priced the
// alternatives to a kernel boundary from the guide's table.  This measures them.)
//
// One "phase" = what a residual-block launch does around its K loop, without the K loop: every workgroup (G of them, one per CU, 256
// threads) reads `kb` KB that OTHER workgroups -- on other XCDs -- wrote in the previous phase, reduces them, and writes its own `kb` KB.
//   mode 0: one kernel launch per phase, N launches recorded in a hipGraph (the product's mechanism)
//   mode 1: ONE persistent launch, a grid-wide barrier between phases (agent-scope arrive counter + spin)
//   mode 2: ONE persistent launch, point-to-point flags: a workgroup waits only for the P workgroups whose data it reads
//   mode 3 / 4: as 1 / 2, but the DATA is written and read with agent-coherent accesses (global_store / global_load ... sc1) and the
//               hand-off is `s_waitcnt vmcnt(0)` + the flag -- no `buffer_wbl2 sc1` / `buffer_inv sc1` (what the agent-scope fences of
//               modes 1 / 2 compile to: a write-back and an invalidate of the whole XCD L2)
// Every mode checks the data it read (phase number baked into the values), so a stale cache line shows up as an error count.
// Spins are bounded: a lost hand-off ends the kernel with an error flag instead of hanging the box.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/handoff.hip -o tools/ubench/handoff && tools/ubench/handoff [G=256] [phases=200] [kb=4] [P=32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kThreads = 256;
constexpr unsigned kMaxSpin = 1u << 22;

__device__ __forceinline__ float value_of(int phase, int block, int i) { return (float)((phase * 131 + block * 7 + i) & 1023); }

// read P producers' slices (kb KB each is split over them), check, write this block's slice for `phase`
template <bool SC1 = false>
__device__ __forceinline__ void phase_body(const float* prev, float* next, int phase, int G, int n_floats, int P,
                                           unsigned* errors) {
  const int b = blockIdx.x;
  unsigned bad = 0;
  if (phase > 0) {
    const int per = n_floats / P;                                  // floats taken from each producer
    for (int i = threadIdx.x; i < n_floats; i += kThreads) {
      const int p = i / per, j = i - p * per;
      const int src = (b + 37 * (p + 1)) % G;                      // other workgroups, spread over the XCDs
      const float v = SC1 ? __hip_atomic_load(prev + (size_t)src * n_floats + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)   // global_load ... sc1
                          : prev[(size_t)src * n_floats + j];
      bad += v != value_of(phase - 1, src, j);
    }
  }
  for (int i = threadIdx.x; i < n_floats; i += kThreads) {
    if (SC1) __hip_atomic_store(next + (size_t)b * n_floats + i, value_of(phase, b, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store ... sc1
    else __builtin_nontemporal_store(value_of(phase, b, i), next + (size_t)b * n_floats + i);
  }
  if (bad) atomicAdd(errors, bad);
}

__global__ __launch_bounds__(kThreads, 1) void k_phase(const float* prev, float* next, int phase, int G, int n_floats, int P, unsigned* errors) {
  phase_body(prev, next, phase, G, n_floats, P, errors);
}

// mode 1 / 2: buffers alternate; ctr counts arrivals (monotonic), flags[b] = last phase block b finished + 1
template <int MODE, bool SC1 = false>
__global__ __launch_bounds__(kThreads, 1) void k_persistent(float* buf0, float* buf1, float* buf2, int phases, int G, int n_floats, int P,
                                                            unsigned* ctr, unsigned* flags, unsigned* errors) {
  const int b = blockIdx.x;
  float* const bufs[3] = {buf0, buf1, buf2};
  for (int ph = 0; ph < phases; ++ph) {
    float* prev = bufs[(ph + 2) % 3];
    float* next = bufs[ph % 3];
    if (ph > 0) {   // acquire: wait for the producers of phase ph - 1
      if (MODE == 1) {
        if (threadIdx.x == 0) {
          unsigned spin = 0;
          while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ph * G && ++spin < kMaxSpin) __builtin_amdgcn_s_sleep(1);
          if (spin >= kMaxSpin) atomicAdd(errors, 1u << 20);
        }
      } else {
        if ((int)threadIdx.x < P) {
          const int src = (b + 37 * ((int)threadIdx.x + 1)) % G;
          unsigned spin = 0;
          while (__hip_atomic_load(flags + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ph && ++spin < kMaxSpin) __builtin_amdgcn_s_sleep(1);
          if (spin >= kMaxSpin) atomicAdd(errors, 1u << 20);
        }
      }
      __syncthreads();
      if (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // invalidate what this CU / XCD may hold of `prev`
    }
    phase_body<SC1>(prev, next, ph, G, n_floats, P, errors);
    if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's write-through stores acknowledged
    __syncthreads();                              // all of this workgroup's stores issued
    if (threadIdx.x == 0) {
      if (!SC1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");    // ... and visible device-wide before the arrival
      if (MODE == 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(flags + b, (unsigned)(ph + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (mode 2: three rotating buffers.  A block in phase ph overwrites what its consumers read in THEIR phase ph - 2; it got here only
    //  after its P producers finished ph - 1, who needed theirs to finish ph - 2 -- two hops of 32 offsets reach every block of the grid.)
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256, phases = argc > 2 ? atoi(argv[2]) : 200, kb = argc > 3 ? atoi(argv[3]) : 4, P = argc > 4 ? atoi(argv[4]) : 32;
  const int n_floats = kb * 256;
  if (n_floats % P || P > kThreads) { printf("kb*256 must be a multiple of P <= 256\n"); return 1; }
  float *b0, *b1, *b2; unsigned *ctr, *flags, *errors;
  CHECK(hipMalloc(&b0, (size_t)G * n_floats * 4)); CHECK(hipMalloc(&b1, (size_t)G * n_floats * 4)); CHECK(hipMalloc(&b2, (size_t)G * n_floats * 4));
  float* bufs[3] = {b0, b1, b2};
  CHECK(hipMalloc(&ctr, 4)); CHECK(hipMalloc(&flags, G * 4)); CHECK(hipMalloc(&errors, 4));
  hipStream_t s; CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("G=%d workgroups x %d threads, %d phases, %d KB per workgroup per phase read from %d producers on other XCDs\n", G, kThreads, phases, kb, P);
  for (int rep = 0; rep < 3; ++rep) {
    for (int mode = 0; mode < 5; ++mode) {
      CHECK(hipMemsetAsync(ctr, 0, 4, s)); CHECK(hipMemsetAsync(flags, 0, G * 4, s)); CHECK(hipMemsetAsync(errors, 0, 4, s));
      float ms = 0;
      if (mode == 0) {
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int ph = 0; ph < phases; ++ph)
          hipLaunchKernelGGL(k_phase, dim3(G), dim3(kThreads), 0, s, bufs[(ph + 2) % 3], bufs[ph % 3], ph, G, n_floats, P, errors);
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHECK(hipGraphLaunch(ge, s)); CHECK(hipStreamSynchronize(s));       // warm
        CHECK(hipMemsetAsync(errors, 0, 4, s));
        CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
      } else {
        for (int w = 0; w < 2; ++w) {                                        // warm, then timed
          CHECK(hipMemsetAsync(ctr, 0, 4, s)); CHECK(hipMemsetAsync(flags, 0, G * 4, s)); CHECK(hipMemsetAsync(errors, 0, 4, s));
          CHECK(hipEventRecord(e0, s));
          if (mode == 1) hipLaunchKernelGGL((k_persistent<1, false>), dim3(G), dim3(kThreads), 0, s, b0, b1, b2, phases, G, n_floats, P, ctr, flags, errors);
          else if (mode == 2) hipLaunchKernelGGL((k_persistent<2, false>), dim3(G), dim3(kThreads), 0, s, b0, b1, b2, phases, G, n_floats, P, ctr, flags, errors);
          else if (mode == 3) hipLaunchKernelGGL((k_persistent<1, true>), dim3(G), dim3(kThreads), 0, s, b0, b1, b2, phases, G, n_floats, P, ctr, flags, errors);
          else hipLaunchKernelGGL((k_persistent<2, true>), dim3(G), dim3(kThreads), 0, s, b0, b1, b2, phases, G, n_floats, P, ctr, flags, errors);
          CHECK(hipEventRecord(e1, s));
          CHECK(hipStreamSynchronize(s));
        }
      }
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned err = 0; CHECK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
      static const char* names[] = {"kernel boundary (hipGraph of launches)", "persistent + grid barrier", "persistent + point-to-point flags",
                                     "grid barrier, sc1 data, no cache fences", "point-to-point flags, sc1 data, no fences"};
      printf("  rep %d  mode %d  %-42s %8.3f us per phase   errors %u%s\n", rep, mode, names[mode], ms * 1e3 / phases, err & 0xfffff,
             (err >> 20) ? "  (SPIN LIMIT HIT)" : "");
    }
  }
  return 0;
}
