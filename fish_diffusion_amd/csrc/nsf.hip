// nsf.hip -- NSF-HiFiGAN generator (fish_diffusion/modules/vocoders/nsf_hifigan/models.py:353-448).
//
//   har  = SourceModuleHnNSF(upsample(f0))                                  VALU + double-precision scans
//   x    = conv_pre(mel)                                                    MFMA conv, k=7
//   per stage i:  x = ConvTranspose1d(lrelu(x)) [polyphase: u*Cout rows, <=5 unit-shift taps]   MFMA conv
//                 x += noise_conv_i(har)                                    VALU (1 input channel, stride)
//                 x = mean_j ResBlock_j(x)   (k = 3,7,11; dil 1,3,5)        MFMA convs, lrelu fused on the B operand,
//                                                                            residual / MRF-mean fused in the epilogue
//   wav  = tanh(conv_post(lrelu(x, 0.01)))                                  VALU (1 output channel)
#include "common.hip.h"

#include <cstdlib>
#include "elementwise.hip.h"
#include "nsf_kernels.hip.h"
#include "convplan.hip.h"
#include "resblock_fused.hip.h"

using namespace fdx;


// ================================================================================================ layout
static int nsf_validate(const fdx_nsf_desc* d) {
  if (!d) return fail(nullptr, FDX_E_ARG, "null nsf desc");
  if (d->n_stages <= 0 || d->n_stages > FDX_MAX_STAGES) return fail(nullptr, FDX_E_ARG, "bad n_stages %d", d->n_stages);
  if (d->n_resblock_kernels <= 0 || d->n_resblock_kernels > FDX_MAX_RESK) return fail(nullptr, FDX_E_ARG, "bad n_resblock_kernels");
  if (d->resblock_type != 1 && d->resblock_type != 2) return fail(nullptr, FDX_E_ARG, "resblock must be \"1\" or \"2\"");
  if (d->n_dilations <= 0 || d->n_dilations > FDX_MAX_DIL) return fail(nullptr, FDX_E_ARG, "bad n_dilations");
  if (d->resblock_type == 2 && d->n_dilations > 2) return fail(nullptr, FDX_E_NOIMPL, "ResBlock2 with more than 2 dilations is not supported");
  if (d->num_mels % 8) return fail(nullptr, FDX_E_ARG, "num_mels must be a multiple of 8");
  int hop = 1;
  for (int i = 0; i < d->n_stages; ++i) {
    const int u = d->upsample_rates[i], k = d->upsample_kernel_sizes[i];
    if (u <= 0 || k < u || ((k - u) & 1)) return fail(nullptr, FDX_E_ARG, "unsupported upsample geometry k=%d u=%d", k, u);
    hop *= u;
    const int c = d->upsample_initial_channel >> (i + 1);
    if (c < 8 || (d->upsample_initial_channel % (1 << (i + 1)))) return fail(nullptr, FDX_E_ARG, "channel count %d at stage %d unsupported", c, i);
  }
  if (hop != d->hop_size) return fail(nullptr, FDX_E_ARG, "hop_size %d != prod(upsample_rates) %d", d->hop_size, hop);
  for (int j = 0; j < d->n_resblock_kernels; ++j) {
    const int k = d->resblock_kernel_sizes[j];
    if (!(k & 1)) return fail(nullptr, FDX_E_ARG, "resblock kernel sizes must be odd");
    for (int q = 0; q < d->n_dilations; ++q)
      if ((k - 1) / 2 * d->resblock_dilations[j][q] > kHalo)
        return fail(nullptr, FDX_E_ARG, "resblock receptive field exceeds the %d-column halo", kHalo);
  }
  if (d->harmonic_num < 0 || d->harmonic_num > 31) return fail(nullptr, FDX_E_ARG, "bad harmonic_num");
  return FDX_OK;
}

// polyphase geometry of ConvTranspose1d(k, stride u, padding (k-u)/2): output n = u*q + r reads input q - delta with
// kernel tap kk = u*delta + r + p.  Returns [delta_lo, delta_hi] over all phases.
static void ups_delta_range(int k, int u, int& lo, int& hi) {
  const int p = (k - u) / 2;
  lo = 1 << 20; hi = -(1 << 20);
  for (int r = 0; r < u; ++r)
    for (int dl = -64; dl <= 64; ++dl) {
      const int kk = u * dl + r + p;
      if (kk >= 0 && kk < k) { lo = dl < lo ? dl : lo; hi = dl > hi ? dl : hi; }
    }
}

static void nsf_layout(const fdx_nsf_desc& d, NsfLayout& l) {
  size_t cur = 0;
  const int H = d.harmonic_num + 1;
  l.src_w = cur; cur += round_up(H, 64);
  l.src_b = cur; cur += 64;
  l.conv_pre = plan_conv(cur, d.upsample_initial_channel, d.num_mels, 7);
  l.stages.clear();
  for (int i = 0; i < d.n_stages; ++i) {
    NsfStage st;
    st.cin = d.upsample_initial_channel >> i;
    st.cout = d.upsample_initial_channel >> (i + 1);
    st.stride = d.upsample_rates[i];
    st.ksize = d.upsample_kernel_sizes[i];
    int lo, hi;
    ups_delta_range(st.ksize, st.stride, lo, hi);
    st.ups = plan_conv(cur, st.stride * st.cout, st.cin, hi - lo + 1);
    st.ups_shift0 = -hi;
    int sprod = 1;
    for (int j = i + 1; j < d.n_stages; ++j) sprod *= d.upsample_rates[j];
    if (i + 1 < d.n_stages) { st.nc_k = 2 * sprod; st.nc_stride = sprod; st.nc_pad = sprod / 2; }
    else { st.nc_k = 1; st.nc_stride = 1; st.nc_pad = 0; }
    st.nc_w = cur; cur += round_up(st.cout * st.nc_k, 64);
    st.nc_b = cur; cur += round_up(st.cout, 64);
    l.stages.push_back(st);
  }
  for (int i = 0; i < d.n_stages; ++i) {
    NsfStage& st = l.stages[i];
    for (int j = 0; j < d.n_resblock_kernels; ++j)
      for (int q = 0; q < d.n_dilations; ++q) {
        st.c1.push_back(plan_conv(cur, st.cout, st.cout, d.resblock_kernel_sizes[j]));
        if (d.resblock_type == 1) st.c2.push_back(plan_conv(cur, st.cout, st.cout, d.resblock_kernel_sizes[j]));
      }
  }
  l.post_c = l.stages.back().cout;
  l.post_w = cur; cur += round_up(l.post_c * 7, 64);
  l.post_b = cur; cur += 64;
  // fused ResBlock1 weights of the small-channel stages (behind everything else: the offsets above do not move)
  for (int i = 0; i < d.n_stages; ++i) {
    NsfStage& st = l.stages[i];
    st.fused = d.resblock_type == 1 && d.n_dilations == kRbPairs && rb_fused_wins(st.cout);
    for (int j = 0; j < d.n_resblock_kernels && st.fused; ++j) {
      // with the CONFIGURED dilations: the six convs' halo has to leave at least 64 owned columns in the LDS window, otherwise conv by conv
      int d1[kRbPairs], d2[kRbPairs];
      for (int q = 0; q < kRbPairs; ++q) { d1[q] = d.resblock_dilations[j][q]; d2[q] = 1; }
      const int H4 = (rb_halo(d.resblock_kernel_sizes[j], d1, d2) + 3) & ~3;
      st.fused = rb_fused_supported(st.cout, d.resblock_kernel_sizes[j]) && rb_pick_n(st.cout, H4, 1L << 20, 1) > 0;
    }
    st.fw.clear(); st.fb.clear();
    if (!st.fused) continue;
    for (int j = 0; j < d.n_resblock_kernels; ++j) {
      st.fw.push_back(cur); cur += round_up((int)(2 * kRbPairs * rb_fused_floats(st.cout, d.resblock_kernel_sizes[j])), 64);
      st.fb.push_back(cur); cur += round_up(2 * kRbPairs * st.cout, 64);
    }
  }
  l.total_floats = cur;
}

extern "C" int fdx_nsf_num_weights(const fdx_nsf_desc* d) {
  if (nsf_validate(d)) return FDX_E_ARG;
  const int per_block = d->n_dilations * (d->resblock_type == 1 ? 4 : 2);
  return 2 + 2 + d->n_stages * 4 + d->n_stages * d->n_resblock_kernels * per_block + 2;
}

extern "C" int fdx_nsf_packed_bytes(const fdx_nsf_desc* d, size_t* bytes) {
  if (nsf_validate(d) || !bytes) return FDX_E_ARG;
  NsfLayout l;
  nsf_layout(*d, l);
  *bytes = l.total_floats * sizeof(float);
  return FDX_OK;
}

extern "C" int fdx_nsf_pack(const fdx_nsf_desc* d, const float* const* w, int n, void* out, size_t bytes) {
  if (nsf_validate(d)) return FDX_E_ARG;
  if (!w || !out) return fail(nullptr, FDX_E_ARG, "null pointer");
  if (n != fdx_nsf_num_weights(d)) return fail(nullptr, FDX_E_ARG, "expected %d weight tensors, got %d", fdx_nsf_num_weights(d), n);
  NsfLayout l;
  nsf_layout(*d, l);
  if (bytes != l.total_floats * sizeof(float)) return fail(nullptr, FDX_E_ARG, "packed size mismatch");
  float* A = static_cast<float*>(out);
  memset(A, 0, bytes);
  const int H = d->harmonic_num + 1;
  int k = 0;
  memcpy(A + l.src_w, w[k], H * sizeof(float));
  A[l.src_b] = w[k + 1][0];
  k += 2;
  pack_conv1d(A, l.conv_pre, w[k], d->upsample_initial_channel, d->num_mels, w[k + 1]);
  k += 2;
  for (int i = 0; i < d->n_stages; ++i) {
    const NsfStage& st = l.stages[i];
    const float* uw = w[k]; const float* ub = w[k + 1];   // ConvTranspose1d weight [cin][cout][ksize]
    const int u = st.stride, p = (st.ksize - u) / 2, delta_hi = -st.ups_shift0;
    const int R = 32 * st.ups.RB;
    pack_convgemm(A + st.ups.w_off, st.ups.n_mtiles, st.ups.RB, st.ups.cin8, st.ups.taps,
                  [&](int mt, int rb, int ii, int c, int tap) -> float {
                    const int row = mt * R + rb * 32 + ii;
                    if (row >= u * st.cout || c >= st.cin) return 0.f;
                    const int r = row / st.cout, co = row % st.cout;
                    const int kk = u * (delta_hi - tap) + r + p;
                    if (kk < 0 || kk >= st.ksize) return 0.f;
                    return uw[((size_t)c * st.cout + co) * st.ksize + kk];
                  });
    for (int c = 0; c < st.cout; ++c) A[st.ups.b_off + c] = ub[c];
    memcpy(A + st.nc_w, w[k + 2], (size_t)st.cout * st.nc_k * sizeof(float));
    memcpy(A + st.nc_b, w[k + 3], (size_t)st.cout * sizeof(float));
    k += 4;
  }
  for (int i = 0; i < d->n_stages; ++i) {
    const NsfStage& st = l.stages[i];
    size_t idx = 0;
    for (int j = 0; j < d->n_resblock_kernels; ++j)
      for (int q = 0; q < d->n_dilations; ++q, ++idx) {
        const int ks = d->resblock_kernel_sizes[j];
        if (st.fused) {     // conv 2q = c1_q, conv 2q + 1 = c2_q
          rb_fused_pack(A + st.fw[j] + (size_t)(2 * q) * rb_fused_floats(st.cout, ks), w[k], st.cout, ks);
          memcpy(A + st.fb[j] + (size_t)(2 * q) * st.cout, w[k + 1], (size_t)st.cout * sizeof(float));
          rb_fused_pack(A + st.fw[j] + (size_t)(2 * q + 1) * rb_fused_floats(st.cout, ks), w[k + 2], st.cout, ks);
          memcpy(A + st.fb[j] + (size_t)(2 * q + 1) * st.cout, w[k + 3], (size_t)st.cout * sizeof(float));
        }
        pack_conv1d(A, st.c1[idx], w[k], st.cout, st.cout, w[k + 1]); k += 2;
        if (d->resblock_type == 1) { pack_conv1d(A, st.c2[idx], w[k], st.cout, st.cout, w[k + 1]); k += 2; }
      }
  }
  memcpy(A + l.post_w, w[k], (size_t)l.post_c * 7 * sizeof(float));
  A[l.post_b] = w[k + 1][0];
  return FDX_OK;
}

extern "C" int fdx_nsf_attach(fdx_handle h, const fdx_nsf_desc* d, const void* dev, size_t bytes) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (nsf_validate(d)) { h->err = g_last_error; return FDX_E_ARG; }
  NsfLayout l;
  nsf_layout(*d, l);
  if (!dev || bytes != l.total_floats * sizeof(float)) return fail(h, FDX_E_ARG, "packed arena size mismatch");
  h->nd = *d; h->nl = l; h->nsf_arena = static_cast<const float*>(dev); h->nsf_ok = true;
  h->vB = h->vT = 0;
  return FDX_OK;
}

// FDX_NSF_FUSED=0: the small-channel stages conv by conv (the path the fused ResBlock1 kernel is tested against)
static bool nsf_fused_enabled() {
  static const bool v = [] { const char* e = getenv("FDX_NSF_FUSED"); return !e || atoi(e) != 0; }();
  return v;
}

// ================================================================================================ noise convs
template <int K>
static void launch_noise_win(float* y, long y_bs, int ldy, const float* har, long har_bs, const float* w, const float* bias, int C,
                             int L, int stride, int pad, int B, hipStream_t s) {
  // enough channel groups to give every CU a few workgroups; each group re-reads its K-sample windows
  const int col_blocks = (L + 255) / 256;
  int groups = 1;
  while (groups < C && (long)col_blocks * groups * B < 1024) groups *= 2;
  const int CG = (C + groups - 1) / groups;
  hipLaunchKernelGGL(k_noise_conv_add_win<K>, dim3(col_blocks, (C + CG - 1) / CG, B), dim3(256), 0, s, y, y_bs, ldy, har, har_bs, w,
                     bias, C, CG, L, stride, pad);
}

static void launch_noise_conv(float* y, long y_bs, int ldy, const float* har, long har_bs, const float* w, const float* bias, int C,
                              int L, int K, int stride, int pad, int B, hipStream_t s) {
  switch (K) {
    case 128: return launch_noise_win<128>(y, y_bs, ldy, har, har_bs, w, bias, C, L, stride, pad, B, s);
    case 64: return launch_noise_win<64>(y, y_bs, ldy, har, har_bs, w, bias, C, L, stride, pad, B, s);
    case 32: return launch_noise_win<32>(y, y_bs, ldy, har, har_bs, w, bias, C, L, stride, pad, B, s);
    case 16: return launch_noise_win<16>(y, y_bs, ldy, har, har_bs, w, bias, C, L, stride, pad, B, s);
    case 8: return launch_noise_win<8>(y, y_bs, ldy, har, har_bs, w, bias, C, L, stride, pad, B, s);
    case 4: return launch_noise_win<4>(y, y_bs, ldy, har, har_bs, w, bias, C, L, stride, pad, B, s);
    case 2: return launch_noise_win<2>(y, y_bs, ldy, har, har_bs, w, bias, C, L, stride, pad, B, s);
    case 1: return launch_noise_win<1>(y, y_bs, ldy, har, har_bs, w, bias, C, L, stride, pad, B, s);
    default:   // any other geometry: one thread per (channel, sample), K-loop over memory
      hipLaunchKernelGGL(k_noise_conv_add, dim3((L + 255) / 256, C, B), dim3(256), 0, s, y, y_bs, ldy, har, har_bs, w, bias, C, L, K,
                         stride, pad);
  }
}

// ================================================================================================ buffers
struct StageGeom { int C, Cp, L, ld; };
static StageGeom stage_geom(const fdx_nsf_desc& d, int T, int i /* -1 = conv_pre output */) {
  StageGeom g;
  int L = T;
  for (int j = 0; j <= i; ++j) L *= d.upsample_rates[j];
  g.C = d.upsample_initial_channel >> (i + 1);
  g.Cp = round_up(g.C, 8);
  g.L = L;
  g.ld = padded_ld(L, 256);
  return g;
}

static int nsf_alloc(fdx_ctx* h, int B, int T, hipStream_t s) {
  const auto& d = h->nd;
  const bool geom = B != h->vB || T != h->vT;
  h->vB = B; h->vT = T;
  const int L = T * d.hop_size, H = d.harmonic_num + 1;
  const int ld0 = padded_ld(T, 256);
  FDX_HIP(h, h->vmel.ensure((size_t)B * d.num_mels * ld0 * 4, geom, s));
  FDX_HIP(h, h->vpre.ensure((size_t)B * d.upsample_initial_channel * ld0 * 4, geom, s));
  FDX_HIP(h, h->vf0up.ensure((size_t)B * L * 4, false, s));
  FDX_HIP(h, h->vscan.ensure((size_t)B * H * L * 4, false, s));
  FDX_HIP(h, h->vhar.ensure((size_t)B * padded_ld(L, 256) * 4, geom, s));
  const int n_chunks = (L + kScanChunk - 1) / kScanChunk;
  FDX_HIP(h, h->scan_part.ensure((size_t)B * H * n_chunks * 8, false, s));
  if ((int)h->vU.size() != d.n_stages) {
    h->vU = std::vector<DevBuf>(d.n_stages); h->vR = std::vector<DevBuf>(d.n_stages);
    h->vTm = std::vector<DevBuf>(d.n_stages); h->vXS = std::vector<DevBuf>(d.n_stages);
  }
  for (int i = 0; i < d.n_stages; ++i) {
    const StageGeom g = stage_geom(d, T, i);
    const size_t bytes = (size_t)B * g.Cp * g.ld * 4;
    FDX_HIP(h, h->vU[i].ensure(bytes, geom, s));
    FDX_HIP(h, h->vR[i].ensure(bytes, geom, s));
    FDX_HIP(h, h->vTm[i].ensure(bytes, geom, s));
    FDX_HIP(h, h->vXS[i].ensure(bytes, geom, s));
  }
  return FDX_OK;
}

// ================================================================================================ source module
static int nsf_source_core(fdx_ctx* h, const float* f0, int B, int T, const float* rand_ini, const float* src_noise,
                           uint64_t seed, float* har, long har_bs, hipStream_t s) {
  const auto& d = h->nd;
  const int L = T * d.hop_size, H = d.harmonic_num + 1;
  const int n_chunks = (L + kScanChunk - 1) / kScanChunk;
  const float sr = (float)d.sampling_rate;
  // random draws: injected (parity) or device Philox (perf)
  if (!rand_ini) {
    FDX_HIP(h, h->scratch_a.ensure((size_t)B * H * 4 + 64, false, s));
    hipLaunchKernelGGL(k_rand_uniform, dim3(1 + B * H / 1024), dim3(256), 0, s, h->scratch_a.f(), (size_t)B * H, seed ^ 0x5eedULL, 0ULL);
    rand_ini = h->scratch_a.f();
  }
  if (!src_noise) {
    const size_t n = (size_t)B * L * H;
    FDX_HIP(h, h->vnoise.ensure(n * 4, false, s));
    hipLaunchKernelGGL(k_randn, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, s, h->vnoise.f(), n, seed, 1ULL << 40);
    src_noise = h->vnoise.f();
  }
  // the fundamental carries no initial phase noise: rand_ini[:, 0] = 0 (models.py:213) -- enforced on a private copy
  FDX_HIP(h, h->scratch_b.ensure((size_t)B * H * 4 + 64, false, s));
  FDX_HIP(h, hipMemcpyAsync(h->scratch_b.p, rand_ini, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
  FDX_HIP(h, hipMemset2DAsync(h->scratch_b.p, (size_t)H * 4, 0, 4, B, s));
  const float* rini = h->scratch_b.f();

  hipLaunchKernelGGL(k_f0_upsample, dim3((L + 255) / 256, B), dim3(256), 0, s, h->vf0up.f(), f0, T, L);
  double* part = reinterpret_cast<double*>(h->scan_part.p);
  const dim3 g3(n_chunks, H, B);
  hipLaunchKernelGGL(k_scan_partial<1>, g3, dim3(kScanThreads), 0, s, part, h->vf0up.f(), (const float*)nullptr, rini, L, H, n_chunks, sr);
  hipLaunchKernelGGL(k_scan_offsets, dim3(B * H), dim3(64), 0, s, part, B * H, n_chunks);
  hipLaunchKernelGGL(k_scan_tmp, g3, dim3(kScanThreads), 0, s, h->vscan.f(), part, h->vf0up.f(), rini, L, H, n_chunks, sr);
  hipLaunchKernelGGL(k_scan_partial<2>, g3, dim3(kScanThreads), 0, s, part, h->vf0up.f(), h->vscan.f(), rini, L, H, n_chunks, sr);
  hipLaunchKernelGGL(k_scan_offsets, dim3(B * H), dim3(64), 0, s, part, B * H, n_chunks);
  hipLaunchKernelGGL(k_source_final, dim3(n_chunks, B), dim3(kScanThreads), 0, s, har, har_bs, part, h->vf0up.f(), h->vscan.f(), rini,
                     src_noise, h->nsf_arena + h->nl.src_w, h->nsf_arena + h->nl.src_b, L, H, n_chunks, sr, 0.1f, 0.003f);
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

extern "C" int fdx_nsf_source(fdx_handle h, const float* f0, int B, int T, const float* rand_ini, const float* src_noise,
                              uint64_t seed, float* har, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->nsf_ok) return fail(h, FDX_E_STATE, "fdx_nsf_source: no weights attached");
  if (!f0 || !har || B <= 0 || T <= 0) return fail(h, FDX_E_ARG, "fdx_nsf_source: bad arguments");
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  if (int rc = nsf_alloc(h, B, T, s)) return rc;
  return nsf_source_core(h, f0, B, T, rand_ini, src_noise, seed, har, (long)T * h->nd.hop_size, s);
}

// ================================================================================================ forward
extern "C" int fdx_nsf_forward(fdx_handle h, const float* mel, const float* f0, int B, int T, float mel_scale,
                               const float* rand_ini, const float* src_noise, uint64_t seed, float* wav, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->nsf_ok) return fail(h, FDX_E_STATE, "fdx_nsf_forward: no weights attached");
  if (!mel || !f0 || !wav || B <= 0 || T <= 0) return fail(h, FDX_E_ARG, "fdx_nsf_forward: bad arguments");
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  if (int rc = nsf_alloc(h, B, T, s)) return rc;
  const auto& d = h->nd;
  const auto& l = h->nl;
  const float* A = h->nsf_arena;
  const int L = T * d.hop_size;
  const int ld0 = padded_ld(T, 256), ldL = padded_ld(L, 256);

  // harmonic source into a zero-haloed row (the noise convs read it with stride and padding)
  float* har = h->vhar.f() + kHalo;
  if (int rc = nsf_source_core(h, f0, B, T, rand_ini, src_noise, seed, har, ldL, s)) return rc;

  // c = mel_scale * mel (nsf_hifigan.py:79-80), staged into the padded layout
  hipLaunchKernelGGL(k_copy_rows, ew_grid(T, B * d.num_mels), dim3(kEwBlock), 0, s, h->vmel.f() + kHalo, (long)d.num_mels * ld0,
                     ld0, mel, (long)d.num_mels * T, T, d.num_mels, T, mel_scale, (const uint8_t*)nullptr);
  const int C0 = d.upsample_initial_channel;
  {
    EpiBias e{};
    e.out = h->vpre.f() + kHalo; e.o_bs = (long)C0 * ld0; e.ldo = ld0; e.bias = A + l.conv_pre.b_off; e.M = C0; e.act = ACT_NONE;
    FDX_HIP(h, (run_conv<false>(A, l.conv_pre, B, T, h->vmel.f() + kHalo, (long)d.num_mels * ld0, ld0, -3, 1, 1.f, e, s)));
  }
  const float* x = h->vpre.f() + kHalo;
  long x_bs = (long)C0 * ld0;
  int x_ld = ld0, x_L = T;
  const int nk = d.n_resblock_kernels, nd = d.n_dilations;
  for (int i = 0; i < d.n_stages; ++i) {
    const NsfStage& st = l.stages[i];
    const StageGeom g = stage_geom(d, T, i);
    const long bs = (long)g.Cp * g.ld;
    float* U = h->vU[i].f() + kHalo; float* R = h->vR[i].f() + kHalo;
    float* Tm = h->vTm[i].f() + kHalo; float* XS = h->vXS[i].f() + kHalo;
    {  // x = ups(lrelu(x, 0.1))
      EpiUps e{};
      e.out = U; e.o_bs = bs; e.ldo = g.ld; e.bias = A + st.ups.b_off; e.Cout = st.cout; e.stride = st.stride; e.Lout = g.L;
      FDX_HIP(h, (run_conv<true>(A, st.ups, B, x_L, x, x_bs, x_ld, st.ups_shift0, 1, 0.1f, e, s)));
    }
    // x = x + noise_convs[i](har_source)
    launch_noise_conv(U, bs, g.ld, har, (long)ldL, A + st.nc_w, A + st.nc_b, st.cout, g.L, st.nc_k, st.nc_stride, st.nc_pad, B, s);
    for (int j = 0; j < nk; ++j) {
      const int k = d.resblock_kernel_sizes[j];
      // small-channel stage: the whole ResBlock1 (six convs) out of LDS in one launch.  The kernel moves 16-byte groups (window origin, L and
      // the row pitch multiples of 4): a geometry that is not (an odd product of upsample rates) takes the per-conv path, not an error.
      if (st.fused && nsf_fused_enabled() && g.L % 4 == 0 && g.ld % 4 == 0) {
        RbFusedArgs fa{};
        fa.X = U; fa.x_bs = bs; fa.ldx = g.ld; fa.out = XS; fa.o_bs = bs; fa.ldo = g.ld;
        fa.W = A + st.fw[j]; fa.bias = A + st.fb[j]; fa.L = g.L;
        for (int q = 0; q < kRbPairs; ++q) { fa.d1[q] = d.resblock_dilations[j][q]; fa.d2[q] = 1; }
        fa.slope = 0.1f;
        fa.mode = (j == 0) ? 0 : (j + 1 == nk ? 2 : 1); fa.div = (float)nk;
        FDX_HIP(h, launch_resblock1_fused(st.cout, k, fa, B, s));
        continue;
      }
      const float* cur = U;   // running x of this resblock
      for (int q = 0; q < nd; ++q) {
        const int dil = d.resblock_dilations[j][q];
        const bool last = (q + 1 == nd);
        // where does this sub-block's output go?  last one accumulates into the MRF mean (models.py:426-432)
        EpiResblock eo{};
        eo.bs = bs; eo.ld = g.ld; eo.M = st.cout; eo.div = (float)nk;
        if (!last) { eo.out = R; eo.mode = 0; }
        else {
          eo.out = XS;
          eo.mode = (j == 0) ? 0 : (j + 1 == nk ? 2 : 1);   // xs = r0; xs += r1; ...; x = xs / nk (nk == 1: xs / 1 == xs)
        }
        if (d.resblock_type == 1) {
          EpiResblock e1{};
          e1.out = Tm; e1.resid = nullptr; e1.bs = bs; e1.ld = g.ld; e1.bias = A + st.c1[j * nd + q].b_off; e1.M = st.cout; e1.mode = 0;
          FDX_HIP(h, (run_conv<true>(A, st.c1[j * nd + q], B, g.L, cur, bs, g.ld, -(k - 1) / 2 * dil, dil, 0.1f, e1, s, &h->prof, PROF_NSF_RESBLOCK)));
          eo.resid = cur; eo.bias = A + st.c2[j * nd + q].b_off;
          FDX_HIP(h, (run_conv<true>(A, st.c2[j * nd + q], B, g.L, Tm, bs, g.ld, -(k - 1) / 2, 1, 0.1f, eo, s, &h->prof, PROF_NSF_RESBLOCK)));
        } else {
          // ResBlock2 (models.py:150-155) activates x IN PLACE (:152): the residual of `xt + x` is leaky_relu(x), and the first iteration
          // rewrites the tensor Generator.forward hands to the next ResBlock2 of the stage (:426-431) -- block j starts from leaky_relu
          // applied j times.  Same values without the extra pass over U: block j's first conv reads U through the slope 0.1^(j+1).
          float slope = 0.1f;
          if (q == 0) for (int r = 0; r < j; ++r) slope *= 0.1f;
          eo.resid = cur; eo.rslope = slope; eo.bias = A + st.c1[j * nd + q].b_off;
          FDX_HIP(h, (run_conv<true>(A, st.c1[j * nd + q], B, g.L, cur, bs, g.ld, -(k - 1) / 2 * dil, dil, slope, eo, s, &h->prof, PROF_NSF_RESBLOCK)));
        }
        cur = R;
      }
    }
    x = XS; x_bs = bs; x_ld = g.ld; x_L = g.L;
  }
  // x = tanh(conv_post(lrelu(x)))  -- F.leaky_relu default slope 0.01 (models.py:434)
  hipLaunchKernelGGL(k_conv_post, dim3((L + 255) / 256, B), dim3(256), 0, s, wav, (long)L, x, x_bs, x_ld, A + l.post_w, A + l.post_b,
                     l.post_c, L, 0.01f);
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}
