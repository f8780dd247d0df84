// core.hip -- handle lifetime, error reporting, profiling hooks and the kernel-level test hook.
#include "common.hip.h"

#include <algorithm>
#include <cstdlib>

using namespace fdx;

namespace fdx { thread_local std::string g_last_error; }

extern "C" int fdx_version(void) { return 100; }

extern "C" int fdx_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n > 0 ? 1 : 0;
}

extern "C" int fdx_create(int device, fdx_handle* out) {
  if (!out) return fail(nullptr, FDX_E_ARG, "fdx_create: null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return fail(nullptr, FDX_E_HIP, "fdx_create: no HIP device visible (%s) -- the HIP path has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(nullptr, FDX_E_ARG, "fdx_create: device %d out of range [0,%d)", device, n);
  hipDeviceProp_t prop;
  FDX_HIP(nullptr, hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, FDX_E_HIP, "fdx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                prop.gcnArchName);
  FDX_HIP(nullptr, hipSetDevice(device));
  fdx_ctx* h = new (std::nothrow) fdx_ctx();
  if (!h) return fail(nullptr, FDX_E_NOMEM, "out of host memory");
  h->device = device;
  if (const char* e = getenv("FDX_NO_GRAPH")) h->use_graphs = !(e[0] && e[0] != '0');
  if (const char* e = getenv("FDX_GRAPH_CACHE")) { const int n = atoi(e); if (n > 0) h->graph_cap = n; }
  *out = h;
  return FDX_OK;
}

extern "C" int fdx_destroy(fdx_handle h) {
  GenScope gen_scope(h);
  if (!h) return FDX_OK;
  (void)hipSetDevice(h->device);
  if (h->rg) fdx_rg_free(h->rg);
  if (h->cn) fdx_cn_free(h->cn);
  if (h->td) fdx_td_free(h->td);
  fdx_mel_free_tables(h);
  for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  for (auto e : h->prof.start) (void)hipEventDestroy(e);
  for (auto e : h->prof.stop) (void)hipEventDestroy(e);
  delete h;
  return FDX_OK;
}

// ------------------------------------------------------------------------------------------------ weights over xGMI, without PyTorch
// SURVEY 8(b) sketched `fdx_bcast_weights(h, rccl_comm, root, stream)`.  The Python host broadcasts the packed arenas with torch.distributed
// (backend "nccl" = RCCL: dist.py::broadcast_model_weights); a host that is NOT PyTorch creates its own communicator (ncclCommInitRank) and
// calls this: ONE ncclBroadcast of the arena's bytes.  RCCL is bound at call time (dlopen) -- the library neither links against it nor needs
// it to load -- and the communicator is the caller's: nothing here owns ranks, rendez-vous or a process group.
#include <dlfcn.h>
extern "C" int fdx_bcast_arena(void* dev_arena, size_t bytes, void* rccl_comm, int root, fdx_stream st) {
  if (!dev_arena || !rccl_comm || root < 0) return fail(nullptr, FDX_E_ARG, "fdx_bcast_arena: null arena / communicator or negative root");
  if (bytes == 0) return FDX_OK;
  typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);   // ncclBroadcast (rccl.h:591); ncclResult_t and ncclDataType_t are ints
  typedef const char* (*err_fn)(int);
  struct Rccl { bcast_fn bcast = nullptr; err_fn errstr = nullptr; std::string why; };
  static const Rccl rccl = [] {                 // (function-local static: bound once, thread-safe)
    Rccl r;
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (lib) {
      r.bcast = reinterpret_cast<bcast_fn>(dlsym(lib, "ncclBroadcast"));
      r.errstr = reinterpret_cast<err_fn>(dlsym(lib, "ncclGetErrorString"));
    }
    if (!r.bcast) { const char* e = dlerror(); r.why = e ? e : "symbol ncclBroadcast missing"; }
    return r;
  }();
  const bcast_fn bcast = rccl.bcast;
  const err_fn errstr = rccl.errstr;
  if (!bcast) return fail(nullptr, FDX_E_NOIMPL, "fdx_bcast_arena: librccl.so (ncclBroadcast) could not be loaded: %s", rccl.why.c_str());
  const int kNcclUint8 = 1;                                                            // rccl.h:460
  const int rc = bcast(dev_arena, dev_arena, bytes, kNcclUint8, root, rccl_comm, as_stream(st));
  if (rc != 0) return fail(nullptr, FDX_E_HIP, "fdx_bcast_arena: ncclBroadcast failed: %s", errstr ? errstr(rc) : "unknown RCCL error");
  return FDX_OK;
}

extern "C" const char* fdx_last_error(fdx_handle h) {
  if (h) return h->err.c_str();
  return g_last_error.c_str();
}

extern "C" int fdx_prof_enable(fdx_handle h, int on) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (on < 0) { h->prof.on = false; return FDX_OK; }   // pause: keep what was recorded for fdx_prof_read
  h->prof.on = on != 0;
  h->prof.stride = on > 1 ? on : 1;   // on = N > 1: sample every N-th launch (keeps the probe effect out of `value`)
  h->prof.seen = 0;
  h->prof.used = 0;
  h->prof.flops_total = 0;
  h->prof.label[0] = 0;
  h->prof.pending[0] = 0;
  h->prof.mixed = false;
  return FDX_OK;
}

extern "C" int fdx_prof_select(fdx_handle h, int kind) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (kind < PROF_WN_CONVGATE || kind >= PROF_KINDS) return fail(h, FDX_E_ARG, "fdx_prof_select: unknown kernel family %d", kind);
  h->prof.kind = kind;
  return FDX_OK;
}

extern "C" int fdx_prof_read(fdx_handle h, int* n_launches, double* total_ms, double* flops_per_launch) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  double tot = 0;
  for (size_t i = 0; i < h->prof.used; ++i) {
    FDX_HIP(h, hipEventSynchronize(h->prof.stop[i]));
    float ms = 0;
    FDX_HIP(h, hipEventElapsedTime(&ms, h->prof.start[i], h->prof.stop[i]));
    tot += ms;
  }
  if (n_launches) *n_launches = (int)h->prof.used;
  if (total_ms) *total_ms = tot;
  if (flops_per_launch) *flops_per_launch = h->prof.used ? h->prof.flops_total / (double)h->prof.used : 0.0;
  h->prof.used = 0;
  h->prof.flops_total = 0;
  return FDX_OK;
}

extern "C" int fdx_prof_label(fdx_handle h, char* buf, size_t cap) {
  if (!h || !buf || !cap) return FDX_E_ARG;
  snprintf(buf, cap, "%s", h->prof.label);
  return FDX_OK;
}

extern "C" int fdx_prof_calibrate(fdx_handle h, fdx_stream st, double* empty_pair_ms) {
  GenScope gen_scope(h);
  if (!h || !empty_pair_ms) return FDX_E_ARG;
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  constexpr int N = 65;
  hipEvent_t a[N], b[N];
  for (int i = 0; i < N; ++i) { FDX_HIP(h, hipEventCreate(&a[i])); FDX_HIP(h, hipEventCreate(&b[i])); }
  for (int i = 0; i < N; ++i) { FDX_HIP(h, hipEventRecord(a[i], s)); FDX_HIP(h, hipEventRecord(b[i], s)); }
  FDX_HIP(h, hipStreamSynchronize(s));
  std::vector<float> ms(N);
  for (int i = 0; i < N; ++i) FDX_HIP(h, hipEventElapsedTime(&ms[i], a[i], b[i]));
  for (int i = 0; i < N; ++i) { (void)hipEventDestroy(a[i]); (void)hipEventDestroy(b[i]); }
  std::nth_element(ms.begin(), ms.begin() + N / 2, ms.end());
  *empty_pair_ms = ms[N / 2];
  return FDX_OK;
}

// y = conv1d(act(x), w, bias, dilation, "same" padding) through the MFMA kernel family -- test hook.
extern "C" int fdx_debug_conv1d(fdx_handle h, const float* x, int B, int Cin, int T, const float* host_w,
                                const float* host_bias, int Cout, int k, int dilation, float in_slope, int mode, float* y,
                                fdx_stream st) {
  GenScope gen_scope(h);
  if (!h || !x || !host_w || !y) return FDX_E_ARG;
  if (B <= 0 || Cin <= 0 || T <= 0 || Cout <= 0 || k <= 0 || !(k & 1) || dilation <= 0)
    return fail(h, FDX_E_ARG, "fdx_debug_conv1d: bad geometry");
  if ((k - 1) / 2 * dilation > kHalo) return fail(h, FDX_E_ARG, "fdx_debug_conv1d: receptive field exceeds the %d-column halo", kHalo);
  if (mode != 0 && mode != 1) return fail(h, FDX_E_ARG, "fdx_debug_conv1d: mode must be 0 or 1");
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  const int cin8 = (Cin + 7) / 8, cinp = cin8 * 8;
  const int n_mtiles = (Cout + 63) / 64;
  const size_t wf = packed_floats(n_mtiles, 2, cin8, k);
  std::vector<float> packed(wf), bias(round_up(Cout, 64), 0.f);
  pack_convgemm(packed.data(), n_mtiles, 2, cin8, k, [&](int mt, int rb, int i, int c, int tap) -> float {
    const int row = mt * 64 + rb * 32 + i;
    if (row >= Cout || c >= Cin) return 0.f;
    return host_w[((size_t)row * Cin + c) * k + tap];
  });
  if (host_bias) memcpy(bias.data(), host_bias, Cout * sizeof(float));
  const int ld = padded_ld(T, 256);
  FDX_HIP(h, h->dbg_w.ensure(wf * 4, false, s));
  FDX_HIP(h, h->dbg_b.ensure(bias.size() * 4, false, s));
  FDX_HIP(h, h->dbg_x.ensure((size_t)B * cinp * ld * 4, true, s));
  FDX_HIP(h, hipMemcpyAsync(h->dbg_w.p, packed.data(), wf * 4, hipMemcpyHostToDevice, s));
  FDX_HIP(h, hipMemcpyAsync(h->dbg_b.p, bias.data(), bias.size() * 4, hipMemcpyHostToDevice, s));
  FDX_HIP(h, hipStreamSynchronize(s));   // host vectors die at return (test hook only)
  for (int b = 0; b < B; ++b)   // item b: rows (b*cinp + c) of the zeroed, padded buffer
    FDX_HIP(h, hipMemcpy2DAsync(h->dbg_x.f() + (size_t)b * cinp * ld + kHalo, (size_t)ld * 4, x + (size_t)b * Cin * T,
                                (size_t)T * 4, (size_t)T * 4, (size_t)Cin, hipMemcpyDeviceToDevice, s));
  EpiBias e{};
  e.out = y; e.o_bs = (long)Cout * T; e.ldo = T; e.bias = h->dbg_b.f(); e.M = Cout; e.act = ACT_NONE; e.tight = 1;
  ConvGeom g{B, T, cin8, k, -(k - 1) / 2 * dilation, dilation, n_mtiles};
  const float4* Wp = reinterpret_cast<const float4*>(h->dbg_w.p);
  const float* X = h->dbg_x.f() + kHalo;
  hipError_t err;
  if (mode == 0) {
    err = in_slope == 1.f ? launch_convgemm<2, true, false, EpiBias>(g, Wp, X, (long)cinp * ld, ld, 1.f, e, s)
                          : launch_convgemm<2, true, true, EpiBias>(g, Wp, X, (long)cinp * ld, ld, in_slope, e, s);
  } else {
    err = in_slope == 1.f ? launch_convgemm<2, false, false, EpiBias>(g, Wp, X, (long)cinp * ld, ld, 1.f, e, s)
                          : launch_convgemm<2, false, true, EpiBias>(g, Wp, X, (long)cinp * ld, ld, in_slope, e, s);
  }
  FDX_HIP(h, err);
  return FDX_OK;
}

#ifdef FDX_KTRACE
// Trace build only (python -m fish_diffusion_amd._build --trace): the next `max_launches` conv/GEMM launches with at most
// `blocks_cap` workgroups write 8 shader-clock stamps per wave into buf[launch][blocks_cap][4][8] (u64, device memory).
extern "C" int fdx_debug_trace(unsigned long long* dev_buf, int max_launches, int blocks_cap) {
  g_trace.buf = dev_buf; g_trace.max_launches = max_launches; g_trace.blocks_cap = blocks_cap; g_trace.n = 0;
  return FDX_OK;
}
#endif
