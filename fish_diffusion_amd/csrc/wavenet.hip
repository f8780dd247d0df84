// wavenet.hip -- WaveNet-residual denoiser + the sampler loop that drives it.
//
// Reference behaviour restated (not translated) from fish_diffusion/modules/wavenet.py:106-120,194-236 and
// archs/diffsinger/diffusions/{diffusion.py:234-311, noise_predictor.py, uni_pc.py:583-818}.
//
// Per denoiser call (B items, T frames, C residual channels, L layers) the device runs
//   1 x  in_proj      GEMM  [C x M]      epilogue: +bias, ReLU, mask, and Y = X + s_0        (wavenet.py:211-218)
//   L x  conv+gate    CONV  [2C x 3C]    epilogue: + hoisted conditioner slab, sigmoid*tanh   (:107-115)
//   L x  out_proj     GEMM  [2C x C]     epilogue: X=(X+res)/sqrt2, Y=X+s_{l+1}, SK(+)=skip   (:117-120,228)
//   1 x  skip_proj    GEMM  [C x C]      epilogue: +bias, ReLU                                 (:229-230)
//   1 x  out_proj     GEMM  [M x C]      epilogue: +bias, mask                                 (:231-234)
// Step-invariant work is hoisted: the L conditioner projections run once per utterance batch
// (fdx_wavenet_prepare) and the step-embedding MLP + L diffusion projections run once per sampler run for
// all timesteps at once.
#include "common.hip.h"
#include "elementwise.hip.h"
#include "convgemm16s.hip.h"
#include "bf16lds.hip.h"
#include "f16s64.hip.h"

#include <cmath>
#include <cstdlib>

using namespace fdx;

// Both residual-block GEMMs run on v_mfma_f32_16x16x4_f32 (convgemm16s.hip.h).  The arena holds the dilated conv's weights in the 16x16x4
// NR = 4 fragment order (pack_convgemm16) and the out-projection's in the generic 32x32x2 order (pack_convgemm), from which its 16x16x4
// orders are derived on the device at attach time.  (Round 1-3 also carried a 32x32x2 residual-block family behind FDX_RESBLOCK_MFMA and
// 128-row MT = 2 tiles behind FDX_MT2_MIN_TILES: measured slower, removed in round 4 -- profiles/NOTES.md.)
// Shape-adaptive tiles for the dilated conv + gate (convgemm16s.hip.h).  FDX_CONV_SHAPE=<NR><NM> (e.g. 27; 44 = the round-1 64 x 64 tile) forces
// one shape: the tile-shape bit-identity test runs every shape in its own process.
static int conv_shape_env() {
  static const int v = [] { const char* e = getenv("FDX_CONV_SHAPE"); return e ? atoi(e) : -1; }();
  return v;
}
// bf16 storage mode: LDS-tiled kernels (bf16lds.hip.h) when a launch has at least this many 128 x 128 tiles; FDX_BF16_LDS=0 disables,
// FDX_BF16_LDS=<n> sets the threshold
static long bf16_lds_min_tiles() {
  static const long v = [] { const char* e = getenv("FDX_BF16_LDS"); return e ? atol(e) : 256L; }();
  return v;
}
static bool bf16_lds_enabled() { return bf16_lds_min_tiles() > 0; }
// fp16-split mode, 128-wide LDS tiles (bf16lds.hip.h, F16S): taken from 200 tiles of 128 x 128 (batch 4 at 10 s); below that the 64 x 64
// tiles of f16s64.hip.h run.  Measured crossover (round 3, ms per 50 UniPC steps, 10 s items; fp32 kernels | 64 x 64 | 128-wide):
// batch 1: 37.1 | 27.2 | 59.4;  2: 66.3 | 40.3 | 66.0;  3: 114.8 | 64.9 | 67.9;  4: 141.8 | 75.0 | 67.9;  6: 208 | 107 | 98.7;  8: 252 | 132 | 104.
// FDX_BF16_LDS overrides this threshold too (the tests force the wide tiles for every geometry with FDX_BF16_LDS=1).
static long f16s_min_tiles() {
  static const long v = [] { const char* e = getenv("FDX_BF16_LDS"); return e ? atol(e) : 200L; }();
  return v;
}
constexpr size_t kBfTileSlack = 8192;   // bytes behind the blocked 16-bit operand buffers (see wn_alloc)
static int outp_shape_env() {   // FDX_OUTP_SHAPE=<NR><NM>: force one out-projection tile shape (NR = 1, 2, 4)
  static const int v = [] { const char* e = getenv("FDX_OUTP_SHAPE"); return e ? atoi(e) : -1; }();
  return v;
}


// ================================================================================================ layout
static int wn_validate(const fdx_wavenet_desc* d) {
  if (!d) return fail(nullptr, FDX_E_ARG, "null wavenet desc");
  if (d->residual_channels <= 0 || d->residual_channels % 32)
    return fail(nullptr, FDX_E_ARG, "residual_channels must be a positive multiple of 32, got %d", d->residual_channels);
  if (d->mel_channels <= 0 || d->mel_channels % 8) return fail(nullptr, FDX_E_ARG, "mel_channels must be a multiple of 8");
  if (d->d_encoder <= 0 || d->d_encoder % 8) return fail(nullptr, FDX_E_ARG, "d_encoder must be a multiple of 8");
  if (d->residual_layers <= 0) return fail(nullptr, FDX_E_ARG, "residual_layers must be positive");
  if (d->dilation_cycle < 0 || d->dilation_cycle > 5)
    return fail(nullptr, FDX_E_ARG, "dilation_cycle %d unsupported (max dilation 16 must fit the %d-column halo)", d->dilation_cycle, kHalo);
  return FDX_OK;
}

static PackedW plan_w(size_t& cur, int rows, int cin, int taps, bool paired, int RB = 2) {
  PackedW p;
  p.RB = RB;
  p.rows = rows;
  p.cin8 = (cin + 7) / 8;
  p.taps = taps;
  p.n_mtiles = paired ? (rows / 2 + 31) / 32 : (rows + 32 * RB - 1) / (32 * RB);
  p.w_off = cur;
  cur += packed_floats(p.n_mtiles, p.RB, p.cin8, p.taps);
  p.b_off = cur;
  cur += (size_t)round_up(rows, 64);
  return p;
}

static void wn_layout(const fdx_wavenet_desc& d, WavenetLayout& l) {
  const int C = d.residual_channels, L = d.residual_layers;
  size_t cur = 0;
  l.in_proj = plan_w(cur, C, d.mel_channels, 1, false, 1);   // (32-row tiles: see skip_proj below)
  l.mlp0 = plan_w(cur, 4 * C, C, 1, false);
  l.mlp2 = plan_w(cur, C, 4 * C, 1, false);
  l.dproj = plan_w(cur, L * C, C, 1, false);
  l.cond = plan_w(cur, L * 2 * C, d.d_encoder, 1, false);
  l.conv.clear(); l.outp.clear(); l.dil.clear();
  for (int i = 0; i < L; ++i) {
    l.conv.push_back(plan_w(cur, 2 * C, C, 3, true));
    l.outp.push_back(plan_w(cur, 2 * C, C, 1, false));
    l.dil.push_back(d.dilation_cycle ? 1 << (i % d.dilation_cycle) : 1);
  }
  // the two once-per-call projections have few rows (C and mel_channels): 32-row tiles double their workgroup count
  // (skip 112 -> 224, out 28 -> 56 at T = 861) where 64-row tiles leave most CUs idle
  l.skip_proj = plan_w(cur, C, C, 1, false, 1);
  l.out_proj = plan_w(cur, d.mel_channels, C, 1, false, 1);
  l.total_floats = cur;
}

extern "C" int fdx_wavenet_num_weights(const fdx_wavenet_desc* d) {
  if (wn_validate(d)) return FDX_E_ARG;
  const int lb = d->use_linear_bias ? 1 : 0;
  return 2 + 2 * (1 + lb) + d->residual_layers * (6 + 1 + lb) + 4;
}

extern "C" int fdx_wavenet_packed_bytes(const fdx_wavenet_desc* d, size_t* bytes) {
  if (wn_validate(d) || !bytes) return FDX_E_ARG;
  WavenetLayout l;
  wn_layout(*d, l);
  *bytes = l.total_floats * sizeof(float);
  return FDX_OK;
}

// plain (unpaired) conv/linear weight [rows][cin][taps] -> fragment order; rows/cin beyond the tensor are 0
static void pack_plain(float* arena, const PackedW& p, const float* w, int rows, int cin, const float* bias) {
  pack_convgemm(arena + p.w_off, p.n_mtiles, p.RB, p.cin8, p.taps, [&](int mt, int rb, int i, int c, int tap) -> float {
    const int row = mt * 32 * p.RB + rb * 32 + i;
    if (row >= rows || c >= cin) return 0.f;
    return w[((size_t)row * cin + c) * p.taps + tap];
  });
  for (int r = 0; r < round_up(rows, 64); ++r) arena[p.b_off + r] = (bias && r < rows) ? bias[r] : 0.f;
}

extern "C" int fdx_wavenet_pack(const fdx_wavenet_desc* d, const float* const* w, int n, void* out, size_t bytes) {
  if (wn_validate(d)) return FDX_E_ARG;
  if (!w || !out) return fail(nullptr, FDX_E_ARG, "null pointer");
  if (n != fdx_wavenet_num_weights(d))
    return fail(nullptr, FDX_E_ARG, "expected %d weight tensors, got %d", fdx_wavenet_num_weights(d), n);
  WavenetLayout l;
  wn_layout(*d, l);
  if (bytes != l.total_floats * sizeof(float)) return fail(nullptr, FDX_E_ARG, "packed size mismatch");
  float* A = static_cast<float*>(out);
  memset(A, 0, bytes);
  const int C = d->residual_channels, L = d->residual_layers, E = d->d_encoder, M = d->mel_channels;
  const int lb = d->use_linear_bias ? 1 : 0;
  int k = 0;
  pack_plain(A, l.in_proj, w[k], C, M, w[k + 1]); k += 2;
  pack_plain(A, l.mlp0, w[k], 4 * C, C, lb ? w[k + 1] : nullptr); k += 1 + lb;
  pack_plain(A, l.mlp2, w[k], C, 4 * C, lb ? w[k + 1] : nullptr); k += 1 + lb;
  std::vector<const float*> dpw(L), dpb(L), cpw(L);
  for (int i = 0; i < L; ++i) {
    const float* conv_w = w[k]; const float* conv_b = w[k + 1]; k += 2;
    dpw[i] = w[k]; dpb[i] = lb ? w[k + 1] : nullptr; k += 1 + lb;
    cpw[i] = w[k]; const float* cp_b = w[k + 1]; k += 2;
    const float* op_w = w[k]; const float* op_b = w[k + 1]; k += 2;
    // dilated conv, gate/filter paired: tile mt holds gate rows 32mt.. (rb 0) and filter rows C+32mt.. (rb 1)
    const PackedW& pc = l.conv[i];
    // 16x16x4 NR = 4 order: rbk 0,1: gate rows 32mt + 16rbk + r;  rbk 2,3: the matching filter rows
    pack_convgemm16(A + pc.w_off, pc.n_mtiles, pc.cin8, 3, [&](int mt, int rbk, int r, int c, int tap) -> float {
      const int ch = mt * 32 + (rbk & 1) * 16 + r;
      if (ch >= C || c >= C) return 0.f;
      return conv_w[((size_t)((rbk >> 1) * C + ch) * C + c) * 3 + tap];
    });
    // the hoisted conditioner slab also absorbs the conv bias: y = (conv + b_conv) + (cond + b_cond), wavenet.py:112
    for (int r = 0; r < 2 * C; ++r) A[l.cond.b_off + (size_t)i * 2 * C + r] = cp_b[r] + conv_b[r];
    pack_plain(A, l.outp[i], op_w, 2 * C, C, op_b);
  }
  {  // diffusion projections of all layers = one [L*C x C] GEMM; conditioner projections = one [L*2C x E] GEMM
    const PackedW& p = l.dproj;
    pack_convgemm(A + p.w_off, p.n_mtiles, 2, p.cin8, 1, [&](int mt, int rb, int r, int c, int) -> float {
      const int row = mt * 64 + rb * 32 + r;
      if (row >= L * C || c >= C) return 0.f;
      return dpw[row / C][(size_t)(row % C) * C + c];
    });
    for (int r = 0; r < L * C; ++r) A[p.b_off + r] = dpb[r / C] ? dpb[r / C][r % C] : 0.f;
    const PackedW& q = l.cond;
    pack_convgemm(A + q.w_off, q.n_mtiles, 2, q.cin8, 1, [&](int mt, int rb, int r, int c, int) -> float {
      const int row = mt * 64 + rb * 32 + r;
      if (row >= L * 2 * C || c >= E) return 0.f;
      return cpw[row / (2 * C)][(size_t)(row % (2 * C)) * E + c];
    });
  }
  pack_plain(A, l.skip_proj, w[k], C, C, w[k + 1]); k += 2;
  pack_plain(A, l.out_proj, w[k], M, C, w[k + 1]); k += 2;
  return FDX_OK;
}

static __global__ void k_acc_vec(float* __restrict__ dst, const float* __restrict__ src, int n) {   // dst += src (attach-time bias sums)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

extern "C" int fdx_wavenet_attach(fdx_handle h, const fdx_wavenet_desc* d, const void* dev, size_t bytes) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (wn_validate(d)) { h->err = g_last_error; return FDX_E_ARG; }
  WavenetLayout l;
  wn_layout(*d, l);
  if (!dev || bytes != l.total_floats * sizeof(float)) return fail(h, FDX_E_ARG, "packed arena size mismatch");
  h->wd = *d;
  h->wl = l;
  h->wn_arena = static_cast<const float*>(dev);
  h->wn_ok = true;
  h->prepared = false;
  ++h->alloc_gen;   // recorded graphs bake the arena (and the derived buffer below) in
  // Shape-adaptive tiles (convgemm16s.hip.h): the dilated conv's weights once more in the NR = 2 fragment order, derived on the
  // device from the packed (NR = 4) arena.  One-off at model load: default stream, synchronous.
  {
    FDX_HIP(h, hipSetDevice(h->device));
    size_t total = 0;
    for (const auto& p : l.conv) total += packed_floats(p.n_mtiles, 2, p.cin8, p.taps);
    FDX_HIP(h, h->wn_nr2.ensure(total * sizeof(float), false, nullptr));
    size_t cur = 0;
    h->wn_nr2_off.clear();
    for (const auto& p : l.conv) {
      const size_t n_src = packed_floats(p.n_mtiles, 2, p.cin8, p.taps) / 4;   // float4 count
      h->wn_nr2_off.push_back(cur);
      hipLaunchKernelGGL(k_repack16_nr2, dim3((unsigned)((n_src + 255) / 256)), dim3(256), 0, nullptr, reinterpret_cast<float2*>(h->wn_nr2.f() + cur),
                         reinterpret_cast<const float4*>(h->wn_arena + p.w_off), n_src, p.cin8 * p.taps, 1);
      cur += n_src * 4;
    }
    // the out-projection (packed in the generic 32x32x2 order) in the three 16x16x4 orders
    {
      size_t tot = 0;
      for (const auto& p : l.outp) tot += 3 * packed_floats(p.n_mtiles, 2, p.cin8, 1);
      FDX_HIP(h, h->wn_outp16.ensure(tot * sizeof(float), false, nullptr));
      h->wn_outp16_off4.clear(); h->wn_outp16_off2.clear(); h->wn_outp16_off1.clear();
      size_t c2 = 0;
      for (const auto& p : l.outp) {
        const size_t nf = packed_floats(p.n_mtiles, 2, p.cin8, 1);
        h->wn_outp16_off4.push_back(c2); h->wn_outp16_off2.push_back(c2 + nf); h->wn_outp16_off1.push_back(c2 + 2 * nf);
        const size_t n4 = nf / 4, n2 = nf / 2;
        hipLaunchKernelGGL(k_repack16_from32<4>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, nullptr, h->wn_outp16.f() + c2, h->wn_arena + p.w_off,
                           p.n_mtiles, p.cin8);
        hipLaunchKernelGGL(k_repack16_from32<2>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, nullptr, h->wn_outp16.f() + c2 + nf, h->wn_arena + p.w_off,
                           p.n_mtiles, p.cin8);
        hipLaunchKernelGGL(k_repack16_from32<1>, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, nullptr, h->wn_outp16.f() + c2 + 2 * nf, h->wn_arena + p.w_off,
                           p.n_mtiles, p.cin8);
        c2 += 3 * nf;
      }
    }
    FDX_HIP(h, hipGetLastError());
    FDX_HIP(h, hipStreamSynchronize(nullptr));
  }
  return FDX_OK;
}

// ================================================================================================ bf16 storage mode (opt-in)
// BASELINE configs[4] ("bf16, 1000-step schedule, batch 128") as SURVEY F4 reads it: bf16 storage, fp32 accumulation.  Only the two
// residual-block GEMMs (94 % of the FLOPs) change: their weights are packed as bf16 fragments of v_mfma_f32_32x32x16_bf16 and
// their activation operands (conv input Y = x + step, gated output Z) are stored C8-blocked bf16; the residual stream, the skip
// sum, the conditioner slab, gates and every accumulation stay fp32.  It cannot meet the fp32 parity bars (bf16 has 8 mantissa
// bits); its error is measured and reported separately (tests) -- never the headline.
struct WnBf16Layout {
  std::vector<size_t> conv, outp;   // offsets in 16-byte units
  size_t total16 = 0;
  int conv_mt = 0, outp_mt = 0, conv_it = 0, outp_it = 0;
};
static void wn_bf16_layout(const fdx_wavenet_desc& d, WnBf16Layout& l) {
  const int C = d.residual_channels, L = d.residual_layers;
  l.conv_mt = C / 32; l.outp_mt = 2 * C / 64; l.conv_it = (C / 16) * 3; l.outp_it = C / 16;
  size_t cur = 0;
  l.conv.clear(); l.outp.clear();
  for (int i = 0; i < L; ++i) {
    l.conv.push_back(cur); cur += (size_t)l.conv_mt * l.conv_it * 2 * 64;
    l.outp.push_back(cur); cur += (size_t)l.outp_mt * l.outp_it * 2 * 64;
  }
  l.total16 = cur;
}
static uint16_t f32_to_bf16(float f) {   // round to nearest even
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));   // inf / nan
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

extern "C" int fdx_wavenet_bf16_packed_bytes(const fdx_wavenet_desc* d, size_t* bytes) {
  if (wn_validate(d) || !bytes) return FDX_E_ARG;
  if (d->residual_channels % 64) return fail(nullptr, FDX_E_ARG, "bf16 mode: residual_channels must be a multiple of 64");
  WnBf16Layout l;
  wn_bf16_layout(*d, l);
  *bytes = l.total16 * 16;
  return FDX_OK;
}

// Same tensor list as fdx_wavenet_pack; only conv_layer.conv.weight and output_projection.conv.weight of each layer are read.
extern "C" int fdx_wavenet_bf16_pack(const fdx_wavenet_desc* d, const float* const* w, int n, void* out, size_t bytes) {
  size_t want = 0;
  if (int rc = fdx_wavenet_bf16_packed_bytes(d, &want)) return rc;
  if (!w || !out || n != fdx_wavenet_num_weights(d) || bytes != want) return fail(nullptr, FDX_E_ARG, "fdx_wavenet_bf16_pack: bad arguments");
  WnBf16Layout l;
  wn_bf16_layout(*d, l);
  uint16_t* A = static_cast<uint16_t*>(out);
  const int C = d->residual_channels, L = d->residual_layers, lb = d->use_linear_bias ? 1 : 0;
  int k = 2 + 2 * (1 + lb);
  for (int i = 0; i < L; ++i) {
    const float* conv_w = w[k];            // [2C][C][3]
    const float* op_w = w[k + 2 + 1 + lb + 2];   // conv(w,b), dproj(w[,b]), cond(w,b), outp(w,b)
    k += 6 + 1 + lb;
    for (int mt = 0; mt < l.conv_mt; ++mt)
      for (int it = 0; it < l.conv_it; ++it) {
        const int cb = it / 3, tap = it % 3;
        for (int rb = 0; rb < 2; ++rb)
          for (int lane = 0; lane < 64; ++lane) {
            uint16_t* dst = A + (l.conv[i] + (((size_t)mt * l.conv_it + it) * 2 + rb) * 64 + lane) * 8;
            const int row = rb * C + mt * 32 + (lane & 31);   // rb 0: gate half, rb 1: filter half of channel mt*32 + i
            for (int j = 0; j < 8; ++j) {
              const int c = cb * 16 + 8 * (lane >> 5) + j;
              dst[j] = f32_to_bf16(conv_w[((size_t)row * C + c) * 3 + tap]);
            }
          }
      }
    for (int mt = 0; mt < l.outp_mt; ++mt)
      for (int it = 0; it < l.outp_it; ++it)
        for (int rb = 0; rb < 2; ++rb)
          for (int lane = 0; lane < 64; ++lane) {
            uint16_t* dst = A + (l.outp[i] + (((size_t)mt * l.outp_it + it) * 2 + rb) * 64 + lane) * 8;
            const int row = mt * 64 + rb * 32 + (lane & 31);
            for (int j = 0; j < 8; ++j) dst[j] = f32_to_bf16(op_w[(size_t)row * C + it * 16 + 8 * (lane >> 5) + j]);
          }
  }
  return FDX_OK;
}

// The bf16 arena derived ON THE DEVICE from the attached fp32 arena -- the same bytes fdx_wavenet_bf16_pack produces on the host
// from the original tensors (round to nearest even of the same fp32 values).  This is what the Python wrapper uses: a rank that
// received its fp32 arena by RCCL broadcast (dist.py) has no meaningful local parameters to pack from.
// One thread per 16-byte group (8 consecutive k of one (row, tap)); `mode16` says which fragment order the fp32 arena holds.
struct Bf16Derive { size_t conv16, outp16; size_t conv_w, outp_w; };   // per layer: bf16 offsets (16-byte units), fp32 offsets (floats)
static __device__ __forceinline__ size_t f32_frag_index32(int mt, int n_it, int it, int rb, int r32, int c) {   // pack_convgemm, RB = 2
  const int hi = (c & 7) >> 2, j = c & 3;
  return ((((size_t)mt * n_it + it) * 2 + rb) * 64 + (hi * 32 + r32)) * 4 + j;
}
static __device__ __forceinline__ size_t f32_frag_index16(int mt, int n_it, int it, int rbk, int r16, int c) {  // pack_convgemm16
  const int h = (c & 7) >> 2, lk = c & 3;
  return ((((size_t)mt * n_it + it) * 2 + h) * 64 + (lk * 16 + r16)) * 4 + rbk;
}
static __global__ void k_bf16_from_arena(__bf16* __restrict__ dst, const float* __restrict__ A, const Bf16Derive* __restrict__ lay, int L,
                                         int C, int conv_mt, int conv_it, int outp_mt, int outp_it, int conv_mode16, int outp_mode16) {
  const size_t per_conv = (size_t)conv_mt * conv_it * 2 * 64, per_outp = (size_t)outp_mt * outp_it * 2 * 64;
  const size_t gidx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= (size_t)L * (per_conv + per_outp)) return;
  const int layer = (int)(gidx / (per_conv + per_outp));
  size_t g = gidx - (size_t)layer * (per_conv + per_outp);
  const Bf16Derive ly = lay[layer];
  bf16x8 v;
  if (g < per_conv) {     // dst16[conv + ((mt*conv_it + it)*2 + rb)*64 + lane], it = cb16*3 + tap: row = rb*C + mt*32 + (lane&31), c = cb16*16 + 8*(lane>>5) + j
    const int lane = (int)(g & 63), rb = (int)((g >> 6) & 1);
    const size_t q = g >> 7;
    const int it = (int)(q % conv_it), mt = (int)(q / conv_it);
    const int cb16 = it / 3, tap = it - cb16 * 3, r32 = lane & 31;
    const int n_it32 = (C / 8) * 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb16 * 16 + 8 * (lane >> 5) + j;
      const int it32 = (c >> 3) * 3 + tap;
      const size_t src = conv_mode16 ? f32_frag_index16(mt, n_it32, it32, (rb << 1) | (r32 >> 4), r32 & 15, c)
                                     : f32_frag_index32(mt, n_it32, it32, rb, r32, c);
      v[j] = (__bf16)A[ly.conv_w + src];
    }
    *reinterpret_cast<bf16x8*>(dst + (ly.conv16 + g) * 8) = v;
  } else {                // out-projection: row = mt*64 + rb*32 + (lane&31), c = it*16 + 8*(lane>>5) + j
    g -= per_conv;
    const int lane = (int)(g & 63), rb = (int)((g >> 6) & 1);
    const size_t q = g >> 7;
    const int it = (int)(q % outp_it), mt = (int)(q / outp_it);
    const int r32 = lane & 31, n_it32 = C / 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = it * 16 + 8 * (lane >> 5) + j;
      const size_t src = outp_mode16 ? f32_frag_index16(mt, n_it32, c >> 3, (rb << 1) | (r32 >> 4), r32 & 15, c)
                                     : f32_frag_index32(mt, n_it32, c >> 3, rb, r32, c);
      v[j] = (__bf16)A[ly.outp_w + src];
    }
    *reinterpret_cast<bf16x8*>(dst + (ly.outp16 + g) * 8) = v;
  }
}

extern "C" int fdx_wavenet_bf16_from_arena(fdx_handle h, void* dev_out, size_t bytes, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->wn_ok) return fail(h, FDX_E_STATE, "fdx_wavenet_bf16_from_arena: attach the fp32 arena first");
  size_t want = 0;
  if (int rc = fdx_wavenet_bf16_packed_bytes(&h->wd, &want)) { h->err = g_last_error; return rc; }
  if (!dev_out || bytes != want) return fail(h, FDX_E_ARG, "fdx_wavenet_bf16_from_arena: output must be %zu bytes", want);
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  WnBf16Layout bl;
  wn_bf16_layout(h->wd, bl);
  const int L = h->wd.residual_layers, C = h->wd.residual_channels;
  std::vector<Bf16Derive> lay(L);
  for (int i = 0; i < L; ++i) lay[i] = Bf16Derive{bl.conv[i], bl.outp[i], h->wl.conv[i].w_off, h->wl.outp[i].w_off};
  FDX_HIP(h, h->scratch_b.ensure(L * sizeof(Bf16Derive), false, s));
  FDX_HIP(h, hipMemcpyAsync(h->scratch_b.p, lay.data(), L * sizeof(Bf16Derive), hipMemcpyHostToDevice, s));
  FDX_HIP(h, hipStreamSynchronize(s));   // `lay` dies at return (one-off set-up call)
  const size_t groups = (size_t)L * ((size_t)bl.conv_mt * bl.conv_it + (size_t)bl.outp_mt * bl.outp_it) * 128;
  hipLaunchKernelGGL(k_bf16_from_arena, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, static_cast<__bf16*>(dev_out), h->wn_arena,
                     static_cast<const Bf16Derive*>(h->scratch_b.p), L, C, bl.conv_mt, bl.conv_it, bl.outp_mt, bl.outp_it, 1, 0);   // arena orders: conv 16x16x4, out-projection 32x32x2
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

// dev_packed == NULL switches the handle back to fp32.  fdx_wavenet_attach must have been called first (biases, the other
// projections and the conditioner slab GEMM come from the fp32 arena).
extern "C" int fdx_wavenet_bf16_attach(fdx_handle h, const void* dev, size_t bytes) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->wn_ok) return fail(h, FDX_E_STATE, "fdx_wavenet_bf16_attach: attach the fp32 arena first");
  if (dev) {
    size_t want = 0;
    if (int rc = fdx_wavenet_bf16_packed_bytes(&h->wd, &want)) { h->err = g_last_error; return rc; }
    if (bytes != want) return fail(h, FDX_E_ARG, "bf16 arena size mismatch");
  }
  if (dev && h->wn_f16s_ok) return fail(h, FDX_E_STATE, "fdx_wavenet_bf16_attach: the fp16-split mode is enabled (disable it first)");
  h->wn_arena_bf16 = dev;
  h->prepared = false;    // the blocked operand buffers are sized in prepare
  ++h->alloc_gen;         // recorded sampler graphs bake the kernel choice in
  // the LDS-tiled kernels' A order (bf16lds.hip.h), a permutation of the same 16-byte groups: one-off, default stream, synchronous
  h->wn_bf16_lds_ok = false;
  bool dil_ok = true;       // the LDS-tiled kernels stage tile +/- 8 columns: layers dilated by more than 8 need the register-direct kernels
  for (int dl : h->wl.dil) dil_ok = dil_ok && dl <= 8;
  if (dev && h->wd.residual_channels % 64 == 0 && bf16_lds_enabled() && dil_ok) {
    FDX_HIP(h, hipSetDevice(h->device));
    WnBf16Layout bl;
    wn_bf16_layout(h->wd, bl);
    const int C = h->wd.residual_channels, L = h->wd.residual_layers;
    FDX_HIP(h, h->wn_bf16_lds.ensure(bl.total16 * 16, false, nullptr));
    const uint4* src = static_cast<const uint4*>(dev);
    uint4* dst = static_cast<uint4*>(h->wn_bf16_lds.p);
    for (int i = 0; i < L; ++i) {
      const size_t nc = (size_t)(C / 64) * (C / 32) * 3 * 512, no = (size_t)(2 * C / 128) * (C / 32) * 512;
      hipLaunchKernelGGL(k_bf16lds_repack, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, nullptr, dst + bl.conv[i], src + bl.conv[i], C / 64, C / 32, 3);
      hipLaunchKernelGGL(k_bf16lds_repack, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, nullptr, dst + bl.outp[i], src + bl.outp[i], 2 * C / 128, C / 32, 1);
    }
    FDX_HIP(h, hipGetLastError());
    FDX_HIP(h, hipStreamSynchronize(nullptr));
    h->wn_bf16_lds_ok = true;
  }
  return FDX_OK;
}

// ================================================================================================ fp16-split mode (opt-in)
// "Past the fp32 roof" (DESIGN section 5): the two residual-block GEMMs with every operand held as a pair of fp16 numbers
// value * 2^k = hi + lo (22 mantissa bits) and each product block formed as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with
// fp32 accumulation: fp32-class results (the dropped lo.lo term and the operands' last two bits are ~2^-22 of a product, below the
// fp32 accumulation error of a K = 512..1536 sum) at a third of the fp16 MFMA rate.  Only the LDS-tiled kernels (bf16lds.hip.h)
// exist for it, so a launch uses it when it has enough tiles (the bf16 mode's threshold) and the fp32 kernels otherwise -- both are
// fp32-class, the mode changes speed, not the contract.  Power-of-two operand scales keep the lo parts out of the fp16 subnormals
// (weights ~0.05 -> * 2^8; gated outputs in (-1, 1) -> * 2^8; conv inputs O(1..100) -> * 2^4) and are removed exactly in the epilogue.
constexpr float kF16sWScale = 256.f, kF16sYScale = 16.f, kF16sZScale = 256.f;
struct F16sDerive { size_t conv_w, outp_w; };   // per layer: fp32 arena offsets (floats)
static size_t f16s_conv_groups(int C) { return (size_t)(C / 64) * (C / 16) * 3 * 4 * 128; }   // 16-byte groups per layer
static size_t f16s_outp_groups(int C) { return (size_t)(2 * C / 128) * (C / 16) * 4 * 128; }
// One thread per 16-byte group of the LDS-order image [m-tile][block16][tap][hl][g][128 rows] (row = wr*64 + x*32 + i):
//   conv (paired rows):  row(x, wr, i) = x*C + (2*mt + wr)*32 + i  (x = 0 gate half, 1 filter half), channels block*16 + 8*g .. +7
//   out-projection:      row = (2*mt + wr)*64 + x*32 + i
static __global__ void k_f16s_from_arena(_Float16* __restrict__ dst, const float* __restrict__ A, const F16sDerive* __restrict__ lay, int L, int C,
                                         int conv_mode16, int outp_mode16, float wscale) {
  const size_t per_conv = (size_t)(C / 64) * (C / 16) * 3 * 4 * 128, per_outp = (size_t)(2 * C / 128) * (C / 16) * 4 * 128;
  const size_t gidx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= (size_t)L * (per_conv + per_outp)) return;
  const int layer = (int)(gidx / (per_conv + per_outp));
  size_t g = gidx - (size_t)layer * (per_conv + per_outp);
  const bool conv = g < per_conv;
  if (!conv) g -= per_conv;
  const int taps = conv ? 3 : 1, n_blk = C / 16;
  const int rho = (int)(g & 127);
  size_t q = g >> 7;
  const int gg = (int)(q & 1); q >>= 1;
  const int hl = (int)(q & 1); q >>= 1;
  const int tap = (int)(q % taps); q /= taps;
  const int blk = (int)(q % n_blk);
  const int mt = (int)(q / n_blk);
  const int wr = rho >> 6, x = (rho >> 5) & 1, i = rho & 31;
  const int mt_old = 2 * mt + wr;                      // the register-direct packings' tile index (32 channels | 64 rows)
  const size_t w_off = conv ? lay[layer].conv_w : lay[layer].outp_w;
  const int n_it32 = conv ? (C / 8) * 3 : C / 8, mode16 = conv ? conv_mode16 : outp_mode16;
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = blk * 16 + 8 * gg + j;
    const int it32 = conv ? (c >> 3) * 3 + tap : (c >> 3);
    const size_t src = mode16 ? f32_frag_index16(mt_old, n_it32, it32, (x << 1) | (i >> 4), i & 15, c) : f32_frag_index32(mt_old, n_it32, it32, x, i, c);
    const float wv = A[w_off + src] * wscale;
    const _Float16 hi = (_Float16)wv;
    v[j] = hl ? (_Float16)(wv - (float)hi) : hi;
  }
  const size_t out = (size_t)layer * (per_conv + per_outp) + (conv ? 0 : per_conv) + g;
  *reinterpret_cast<f16x8*>(dst + out * 8) = v;
}

// The same weights in the small-tile kernel's order (f16s64.hip.h): [m-tile64][block32][tap][hl][k-group 4][64 rows], row = wr*32 + x*16 + i:
//   conv (paired rows):  channel mt*32 + wr*16 + i, row x*C + channel (x = 0 gate half, 1 filter half);  out-projection: row mt*64 + r.
static size_t f16s64_conv_groups(int C) { return (size_t)(2 * C / 64) * (C / 32) * 3 * 2 * 4 * 64; }
static size_t f16s64_outp_groups(int C) { return (size_t)(2 * C / 64) * (C / 32) * 2 * 4 * 64; }
static __global__ void k_f16s64_from_arena(_Float16* __restrict__ dst, const float* __restrict__ A, const F16sDerive* __restrict__ lay, int L, int C,
                                           int conv_mode16, int outp_mode16, float wscale) {
  const size_t per_conv = (size_t)(2 * C / 64) * (C / 32) * 3 * 2 * 4 * 64, per_outp = (size_t)(2 * C / 64) * (C / 32) * 2 * 4 * 64;
  const size_t gidx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= (size_t)L * (per_conv + per_outp)) return;
  const int layer = (int)(gidx / (per_conv + per_outp));
  size_t g = gidx - (size_t)layer * (per_conv + per_outp);
  const bool conv = g < per_conv;
  if (!conv) g -= per_conv;
  const int taps = conv ? 3 : 1, n_blk = C / 32;
  const int r = (int)(g & 63);
  size_t q = g >> 6;
  const int kg = (int)(q & 3); q >>= 2;
  const int hl = (int)(q & 1); q >>= 1;
  const int tap = (int)(q % taps); q /= taps;
  const int blk = (int)(q % n_blk);
  const int mt = (int)(q / n_blk);
  const int wr = r >> 5, x = (r >> 4) & 1, i = r & 15;
  // the register-direct packings address a row as (tile, 32-row half rb, row r32 within it): conv tiles are 32 channels x {gate, filter}
  const int rb = conv ? x : wr, r32 = conv ? wr * 16 + i : x * 16 + i;
  const size_t w_off = conv ? lay[layer].conv_w : lay[layer].outp_w;
  const int n_it32 = conv ? (C / 8) * 3 : C / 8, mode16 = conv ? conv_mode16 : outp_mode16;
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = blk * 32 + 8 * kg + j;
    const int it32 = conv ? (c >> 3) * 3 + tap : (c >> 3);
    const size_t src = mode16 ? f32_frag_index16(mt, n_it32, it32, (rb << 1) | (r32 >> 4), r32 & 15, c) : f32_frag_index32(mt, n_it32, it32, rb, r32, c);
    const float wv = A[w_off + src] * wscale;
    const _Float16 hi = (_Float16)wv;
    v[j] = hl ? (_Float16)(wv - (float)hi) : hi;
  }
  const size_t out = (size_t)layer * (per_conv + per_outp) + (conv ? 0 : per_conv) + g;
  *reinterpret_cast<f16x8*>(dst + out * 8) = v;
}
// The small-tile image is derived alongside and f16s64_kernel serves every launch below the wide-tile threshold: it is never slower than
// the fp32 kernels (1 x 1.25 s: 23.5 vs 23.8 ms per 50 steps -- both launch-bound; 1 x 5 s: 24.8 vs 25.4; 1 x 7.5 s: 26.5 vs 33.4; 1 x 10 s:
// 27.2 vs 37.1).  FDX_F16S_SMALL=0 switches it off (the fp32 kernels below the wide-tile threshold, round 2's behaviour); =<n>: only
// launches with at least n 64 x 64 tiles.
static long f16s_small_min_tiles() {
  static const long v = [] { const char* e = getenv("FDX_F16S_SMALL"); return e ? atol(e) : 1L; }();
  return v;
}

// on != 0: derive the {hi, lo} weights from the attached fp32 arena and use the fp16-split kernels where a launch has enough tiles;
// on == 0: back to the fp32 kernels everywhere.  (Mutually exclusive with the bf16 storage mode: attach that one with NULL first.)
extern "C" int fdx_wavenet_f16s_enable(fdx_handle h, int on) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->wn_ok) return fail(h, FDX_E_STATE, "fdx_wavenet_f16s_enable: attach the fp32 arena first");
  h->wn_f16s_ok = false;
  h->prepared = false;    // the blocked operand buffers are sized in prepare
  ++h->alloc_gen;         // recorded sampler graphs bake the kernel choice in
  if (!on) return FDX_OK;
  if (h->wn_arena_bf16) return fail(h, FDX_E_STATE, "fdx_wavenet_f16s_enable: the bf16 storage mode is attached (detach it first)");
  const int C = h->wd.residual_channels, L = h->wd.residual_layers;
  if (C % 64) return FDX_OK;   // no LDS tiling for this width: the fp32 kernels serve every launch (same fp32-class contract)
  // both fp16-split kernel families stage the conv's window as tile +/- 8 columns (f16s64.hip.h PAD, bf16lds.hip.h): a dilation above 8
  // would read outside the staged window
  for (int dl : h->wl.dil)
    if (dl > 8) return fail(h, FDX_E_NOIMPL, "fdx_wavenet_f16s_enable: dilation %d exceeds the 8-column window pad of the fp16-split kernels (dilation_cycle <= 4)", dl);
  FDX_HIP(h, hipSetDevice(h->device));
  const size_t groups = (size_t)L * (f16s_conv_groups(C) + f16s_outp_groups(C));
  FDX_HIP(h, h->wn_f16s.ensure(groups * 16, false, nullptr));
  std::vector<F16sDerive> lay(L);
  for (int i = 0; i < L; ++i) lay[i] = F16sDerive{h->wl.conv[i].w_off, h->wl.outp[i].w_off};
  FDX_HIP(h, h->scratch_b.ensure(L * sizeof(F16sDerive), false, nullptr));
  FDX_HIP(h, hipMemcpy(h->scratch_b.p, lay.data(), L * sizeof(F16sDerive), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_f16s_from_arena, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, nullptr, static_cast<_Float16*>(h->wn_f16s.p), h->wn_arena,
                     static_cast<const F16sDerive*>(h->scratch_b.p), L, C, 1, 0, kF16sWScale);
  FDX_HIP(h, hipGetLastError());
  h->wn_f16s64_ok = false;
  if (f16s_small_min_tiles() > 0) {
    const size_t g64 = (size_t)L * (f16s64_conv_groups(C) + f16s64_outp_groups(C));
    FDX_HIP(h, h->wn_f16s64.ensure(g64 * 16, false, nullptr));
    hipLaunchKernelGGL(k_f16s64_from_arena, dim3((unsigned)((g64 + 255) / 256)), dim3(256), 0, nullptr, static_cast<_Float16*>(h->wn_f16s64.p), h->wn_arena,
                       static_cast<const F16sDerive*>(h->scratch_b.p), L, C, 1, 0, kF16sWScale);
    FDX_HIP(h, hipGetLastError());
    h->wn_f16s64_ok = true;
  }
  FDX_HIP(h, hipStreamSynchronize(nullptr));
  h->wn_f16s_ok = true;
  return FDX_OK;
}

// Y [B][C][ld] fp32 -> blocked {hi, lo} fp16 (bf_store_quad's layout), scaled: the first layer's conv input
static __global__ void k_to_blocked_f16s(_Float16* __restrict__ dst, long d_bs, const float* __restrict__ src, long s_bs, int ld, int C, int T, float scale) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int b = blockIdx.y / (C / 8), cb = blockIdx.y - b * (C / 8);
  f16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = src[b * s_bs + (long)(cb * 8 + j) * ld + t] * scale;
    hi[j] = (_Float16)v;
    lo[j] = (_Float16)(v - (float)hi[j]);
  }
  _Float16* p = dst + b * d_bs + ((long)((cb >> 1) * 4 + (cb & 1)) * ld + t) * 8;
  *reinterpret_cast<f16x8*>(p) = hi;
  *reinterpret_cast<f16x8*>(p + (long)2 * ld * 8) = lo;
}

// Y [B][C][ld] fp32 -> C8-blocked bf16 (the input projection's second output, once per denoiser call)
static __global__ void k_to_blocked_bf16(__bf16* __restrict__ dst, long d_bs, const float* __restrict__ src, long s_bs, int ld, int C, int T) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const int b = blockIdx.y / (C / 8), cb = blockIdx.y - b * (C / 8);
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)src[b * s_bs + (long)(cb * 8 + j) * ld + t];
  *reinterpret_cast<bf16x8*>(dst + b * d_bs + ((long)cb * ld + t) * 8) = v;
}

// ================================================================================================ launch helpers
// (An 8-wave variant -- 2 K-splitting waves per SIMD, convgemm_kernel<..., NW = 8> -- was measured on the batch-1
// denoiser: 2 % SLOWER than 4 waves.  The K loop's residual stall is per-CU operand throughput, not latency, so a second
// wave per SIMD has nothing to hide)
template <bool SPLITK, bool LRELU, class Epi>
static hipError_t run_gemm(const float* arena, const PackedW& p, int B, int T, const float* X, long x_bs, int ldx,
                           int shift0, int dshift, float slope, const Epi& e, hipStream_t s, hipEvent_t ev0 = nullptr,
                           hipEvent_t ev1 = nullptr) {
  ConvGeom g{B, T, p.cin8, p.taps, shift0, dshift, p.n_mtiles};
  const float4* Wp = reinterpret_cast<const float4*>(arena + p.w_off);
  if constexpr (!Epi::kPaired)
    if (p.RB == 1) return launch_convgemm<1, SPLITK, LRELU, Epi>(g, Wp, X, x_bs, ldx, slope, e, s, ev0, ev1);
  return launch_convgemm<2, SPLITK, LRELU, Epi>(g, Wp, X, x_bs, ldx, slope, e, s, ev0, ev1);
}

static EpiBias epi_bias(float* out, long o_bs, int ldo, const float* bias, int M, int act) {
  EpiBias e{};
  e.out = out; e.o_bs = o_bs; e.ldo = ldo; e.bias = bias; e.M = M; e.act = act;
  return e;
}

// ================================================================================================ prepare
static int wn_alloc(fdx_ctx* h, int B, int T, hipStream_t s) {
  const auto& d = h->wd;
  const int C = d.residual_channels, L = d.residual_layers, M = d.mel_channels, E = d.d_encoder;
  // rows: kHalo zeros, T values, zeros up to the next multiple of 64 + kTailPad (the shape-adaptive tiles may overhang by up to
  // 127 columns, a dilated tap reads 16 further)
  const int ld = kHalo + round_up(T, 64) + kTailPad;
  const bool geom = (B != h->B || T != h->T || h->den_kind != 0);
  h->B = B; h->T = T; h->ld = ld;
  {  // tile shape of the dilated conv + gate for this geometry
    const int rows16 = 2 * C / 16, n_per_wave = (C / 8 * 3 + 3) / 4;
    Shape16 sh{4, 4};
    const long wg44 = (long)(rows16 / 4) * B * ((T + 63) / 64);
    if (wg44 < 2 * 256) sh = pick_shape16(rows16, B, T, 12000.0 / (32.0 * n_per_wave));
    const int forced = conv_shape_env();
    if (forced > 0) sh = Shape16{forced / 10, forced % 10};
    if ((sh.NR != 2 && sh.NR != 4) || sh.NM < 4 || sh.NM > 8) sh = Shape16{4, 4};
    h->conv_shape_nr = sh.NR; h->conv_shape_nm = sh.NM;
    // the out-projection (rows = 2C, K = C: 16 iterations per K-splitting wave) runs on the same 16x16x4 family for EVERY geometry,
    // so that an item's result does not depend on the batch it rode in
    Shape16 so{4, 4};
    {
      const int forced_o = outp_shape_env();
      const int n_o = (C / 8 + 3) / 4;
      const long wgo = (long)(rows16 / 4) * B * ((T + 63) / 64);
      if (forced_o > 0) so = Shape16{forced_o / 10, forced_o % 10};
      else if (wgo < 2 * 256) so = pick_shape16(rows16, B, T, 12000.0 / (32.0 * n_o));
      if ((so.NR != 1 && so.NR != 2 && so.NR != 4) || so.NM < 4 || so.NM > 8) so = Shape16{4, 4};
    }
    h->outp_shape_nr = so.NR; h->outp_shape_nm = so.NM;
  }
  auto sz = [&](int ch) { return (size_t)B * ch * ld * sizeof(float); };
  // every buffer that is read with column shifts must have zero halos => re-zero on geometry change
  FDX_HIP(h, h->xin.ensure(sz(M), geom, s));
  FDX_HIP(h, h->X.ensure(sz(C), geom, s));
  FDX_HIP(h, h->Y.ensure(sz(C), geom, s));
  FDX_HIP(h, h->Z.ensure(sz(C), geom, s));
  FDX_HIP(h, h->SK.ensure(sz(C), geom, s));
  FDX_HIP(h, h->H.ensure(sz(C), geom, s));
  FDX_HIP(h, h->EPS.ensure(sz(M), geom, s));
  FDX_HIP(h, h->condp.ensure(sz(E), geom, s));
  FDX_HIP(h, h->P.ensure(sz(L * 2 * C), geom, s));
  if (h->wn_arena_bf16) {   // blocked bf16 operands: same columns (and zero halos) as the fp32 rows, 2 bytes per element
    // (their own geometry stamp: the geometry may have changed while the handle ran in fp32 mode, which does not touch them)
    const bool gb = B != h->bf16_B || T != h->bf16_T;
    // (+ slack: the LDS kernels' 256-column tiles read their whole window, i.e. up to 100 columns past a short row's end -- into
    // the next row, whose columns feed only outputs >= T; after the last row that is past the buffer)
    FDX_HIP(h, h->Yb.ensure(sz(C) / 2 + kBfTileSlack, gb, s));
    FDX_HIP(h, h->Zb.ensure(sz(C) / 2 + kBfTileSlack, gb, s));
    h->bf16_B = B; h->bf16_T = T;
  }
  if (h->wn_f16s_ok) {      // blocked {hi, lo} fp16 operands: 2 x 2 bytes per element
    const bool gf = B != h->f16s_B || T != h->f16s_T;
    FDX_HIP(h, h->Yh.ensure(sz(C) + kBfTileSlack, gf, s));
    FDX_HIP(h, h->Zh.ensure(sz(C) + kBfTileSlack, gf, s));
    h->f16s_B = B; h->f16s_T = T;
  }
  return FDX_OK;
}

// P[b][l*2C + r][t] = Wc_l[r] . cond[b][:, t] + bc_l[r] + bconv_l[r]   -- all layers in one GEMM (wavenet.py:108,112)
static int wn_cond_slab(fdx_ctx* h, const float* condp, float* P, hipStream_t s) {
  const auto& d = h->wd;
  const int C = d.residual_channels, L = d.residual_layers, E = d.d_encoder, ld = h->ld;
  EpiBias e = epi_bias(P + kHalo, (long)L * 2 * C * ld, ld, h->wn_arena + h->wl.cond.b_off, L * 2 * C, ACT_NONE);
  FDX_HIP(h, (run_gemm<true, false>(h->wn_arena, h->wl.cond, h->B, h->T, condp + kHalo, (long)E * ld, ld, 0, 0, 1.f, e, s)));
  return FDX_OK;
}

extern "C" int fdx_wavenet_prepare(fdx_handle h, const float* cond, int B, int T, const uint8_t* cond_mask, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->wn_ok) return fail(h, FDX_E_STATE, "fdx_wavenet_prepare: no weights attached");
  if (!cond || B <= 0 || T <= 0) return fail(h, FDX_E_ARG, "fdx_wavenet_prepare: bad cond/B/T");
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  if (int rc = wn_alloc(h, B, T, s)) return rc;
  const auto& d = h->wd;
  const int E = d.d_encoder, ld = h->ld;
  // conditioner.masked_fill(cond_masks) (wavenet.py:220-221) while staging into the padded layout
  hipLaunchKernelGGL(k_copy_rows, ew_grid(T, B * E), dim3(kEwBlock), 0, s, h->condp.f() + kHalo, (long)E * ld, ld, cond,
                     (long)E * T, T, E, T, 1.f, cond_mask);
  if (int rc = wn_cond_slab(h, h->condp.f(), h->P.f(), s)) return rc;
  // PLMS evaluates the denoiser once WITHOUT masks (diffusion.py:285): keep the unmasked conditioner for that call
  h->den_kind = 0; h->den_M = d.mel_channels;
  h->cond_masked = cond_mask != nullptr;
  if (h->cond_masked) {
    const bool geom = h->condraw.cap < (size_t)B * E * ld * sizeof(float) || h->condraw_ld != ld;
    FDX_HIP(h, h->condraw.ensure((size_t)B * E * ld * sizeof(float), geom, s));
    h->condraw_ld = ld;
    hipLaunchKernelGGL(k_copy_rows, ew_grid(T, B * E), dim3(kEwBlock), 0, s, h->condraw.f() + kHalo, (long)E * ld, ld, cond,
                       (long)E * T, T, E, T, 1.f, (const uint8_t*)nullptr);
  }
  h->prepared = true;
  return FDX_OK;
}

// ================================================================================================ step embeddings
// S[(l*C + c)][j] = diffusion_projection_l(mlp(embedding(t_j)))[c] for j < n   (wavenet.py:214-215,107)
static int wn_embed(fdx_ctx* h, const float* t_dev, int n, hipStream_t s) {
  const auto& d = h->wd;
  const int C = d.residual_channels, L = d.residual_layers;
  const int ldn = padded_ld(n, 64);
  const bool geom = ldn != h->ldn;
  h->ldn = ldn; h->n_emb = n;
  FDX_HIP(h, h->E.ensure((size_t)C * ldn * 4, geom, s));
  FDX_HIP(h, h->Hm.ensure((size_t)4 * C * ldn * 4, geom, s));
  FDX_HIP(h, h->S0.ensure((size_t)C * ldn * 4, geom, s));
  FDX_HIP(h, h->S.ensure((size_t)L * C * ldn * 4, geom, s));
  const float* A = h->wn_arena;
  hipLaunchKernelGGL(k_step_embed, ew_grid(n, C), dim3(kEwBlock), 0, s, h->E.f() + kHalo, ldn, t_dev, n, C);
  EpiBias e0 = epi_bias(h->Hm.f() + kHalo, 0, ldn, A + h->wl.mlp0.b_off, 4 * C, ACT_MISH);
  FDX_HIP(h, (run_gemm<true, false>(A, h->wl.mlp0, 1, n, h->E.f() + kHalo, 0, ldn, 0, 0, 1.f, e0, s)));
  EpiBias e1 = epi_bias(h->S0.f() + kHalo, 0, ldn, A + h->wl.mlp2.b_off, C, ACT_NONE);
  FDX_HIP(h, (run_gemm<true, false>(A, h->wl.mlp2, 1, n, h->Hm.f() + kHalo, 0, ldn, 0, 0, 1.f, e1, s)));
  EpiBias e2 = epi_bias(h->S.f() + kHalo, 0, ldn, A + h->wl.dproj.b_off, L * C, ACT_NONE);
  FDX_HIP(h, (run_gemm<true, false>(A, h->wl.dproj, 1, n, h->S0.f() + kHalo, 0, ldn, 0, 0, 1.f, e2, s)));
  return FDX_OK;
}

// ================================================================================================ forward
// xin: padded [B][M][ld] (valid data at +kHalo).  Step projections are read from column `col0 + b*sb_bs` of S.
// eps_out: [B][M] rows with pitch ldo / item stride o_bs (padded EPS buffer or the caller's tensor).
// `fuse` (UniPC sampler only): the last projection's epilogue applies the corrector (+ the next step's predictor) to eps instead of storing it
static int wn_forward_core(fdx_ctx* h, const float* xin, int col0, int sb_bs, const uint8_t* mask, float* eps_out,
                           long o_bs, int ldo, hipStream_t s, const float* Pslab = nullptr, const EpiUniPC* fuse = nullptr) {
  const auto& d = h->wd;
  const auto& l = h->wl;
  const int C = d.residual_channels, L = d.residual_layers, M = d.mel_channels;
  const int B = h->B, T = h->T, ld = h->ld, ldn = h->ldn;
  const float* A = h->wn_arena;
  const long bsC = (long)C * ld;
  float* X = h->X.f() + kHalo; float* Y = h->Y.f() + kHalo; float* Z = h->Z.f() + kHalo;
  float* SK = h->SK.f() + kHalo; float* H = h->H.f() + kHalo;
  const float* S = h->S.f() + kHalo + col0;
  const float* keep = h->ragged_keep;   // exact-mask mode (fdx_sampler_run_ragged), else null
  if (keep && h->wn_arena_bf16) return fail(h, FDX_E_NOIMPL, "exact-mask runs are built for the fp32 / fp16-split kernels (not bf16 storage)");

  {  // input projection + ReLU + mask; Y = X + s_0
    EpiBias e = epi_bias(X, bsC, ld, A + l.in_proj.b_off, C, ACT_RELU);
    e.mask = mask; e.mask_ld = T;
    e.out2 = Y; e.o2_bs = bsC; e.ldo2 = ld; e.sb = S; e.sb_ld = ldn; e.sb_bs = sb_bs;
    e.out2_zero_masked = keep != nullptr;   // exact-mask mode: the conv's input is 0 where no frame exists (as if every item ran alone)
    FDX_HIP(h, (run_gemm<true, false>(A, l.in_proj, B, T, xin, (long)M * ld, ld, 0, 0, 1.f, e, s)));
  }
  if (h->wn_arena_bf16)
    hipLaunchKernelGGL(k_to_blocked_bf16, ew_grid(T, B * (C / 8)), dim3(kEwBlock), 0, s, reinterpret_cast<__bf16*>(h->Yb.p) + (size_t)kHalo * 8,
                       (long)C * ld, Y, bsC, ld, C, T);
  // fp16-split mode: taken per call when the launches have enough LDS tiles (else the fp32 kernels: both are fp32-class)
  const bool f16s_big = h->wn_f16s_ok && !h->wn_arena_bf16 && f16s_min_tiles() > 0 && (long)B * ((T + 127) / 128) * (C / 64) >= f16s_min_tiles();
  const bool f16s_small = !f16s_big && h->wn_f16s_ok && h->wn_f16s64_ok && !h->wn_arena_bf16 && (long)B * ((T + 63) / 64) * (C / 32) >= f16s_small_min_tiles();
  const bool f16s = f16s_big || f16s_small;
  _Float16* Yh = f16s ? reinterpret_cast<_Float16*>(h->Yh.p) + (size_t)kHalo * 8 : nullptr;
  _Float16* Zh = f16s ? reinterpret_cast<_Float16*>(h->Zh.p) + (size_t)kHalo * 8 : nullptr;
  const long bsH = (long)2 * C * ld;             // fp16 elements per item ({hi, lo})
  if (f16s)
    hipLaunchKernelGGL(k_to_blocked_f16s, ew_grid(T, B * (C / 8)), dim3(kEwBlock), 0, s, Yh, bsH, Y, bsC, ld, C, T, kF16sYScale);
  const float sqrtL = (float)std::sqrt((double)L);
  if (h->prof.on) {   // fdx_prof_label: the instantiations this call's two residual-block launches are (rocprofv3's kernel names)
    if (f16s_small) {
      h->prof.note(PROF_WN_CONVGATE, "f16s64_kernel<BfEpiGate> (3 x v_mfma_f32_16x16x32_f16 per product block; 64 x 64 workgroup tile, LDS-DMA; %ld workgroups)",
                   (long)B * ((T + 63) / 64) * (2 * C / 64));
      h->prof.note(PROF_WN_OUTPROJ, "f16s64_kernel<BfEpiResSkip> (3 x v_mfma_f32_16x16x32_f16 per product block; 64 x 64 workgroup tile)");
    } else if (f16s_big || (h->wn_arena_bf16 && h->wn_bf16_lds_ok && (long)B * ((T + 127) / 128) * (C / 64) >= bf16_lds_min_tiles())) {
      const int wn = bf16lds_pick_wn(B, T, 2 * C);
      const char* mf = f16s_big ? "3 x v_mfma_f32_32x32x16_f16 per product block" : "v_mfma_f32_32x32x16_bf16";
      h->prof.note(PROF_WN_CONVGATE, "bf16lds_kernel<BfEpiGate, %d, %d> (%s; 128 x %d workgroup tile, operands into LDS by DMA)", wn, (int)f16s_big, mf, 64 * wn);
      h->prof.note(PROF_WN_OUTPROJ, "bf16lds_kernel<BfEpiResSkip, %d, %d> (%s; 128 x %d workgroup tile)", wn, (int)f16s_big, mf, 64 * wn);
    } else if (h->wn_arena_bf16) {
      h->prof.note(PROF_WN_CONVGATE, "convgemm_kernel<2, true, 0, EpiGateB, OPK_BF16> (v_mfma_f32_32x32x16_bf16; 64 x 64 split-K workgroup tile, register-direct operands)");
      h->prof.note(PROF_WN_OUTPROJ, "convgemm_kernel<2, true, 0, EpiResSkipB, OPK_BF16> (v_mfma_f32_32x32x16_bf16; 64 x 64 split-K workgroup tile)");
    } else {
      if (h->conv_shape_nr == 4 && h->conv_shape_nm == 4)
        h->prof.note(PROF_WN_CONVGATE, "convgemm16_kernel<EpiGate16> (v_mfma_f32_16x16x4_f32; 64 x 64 split-K workgroup tile)");
      else
        h->prof.note(PROF_WN_CONVGATE, "convgemm16s_kernel<EpiGate16S<%d>, %d, %d> (v_mfma_f32_16x16x4_f32; %d x %d split-K workgroup tile, %ld workgroups)",
                     h->conv_shape_nm, h->conv_shape_nr, h->conv_shape_nm, 16 * h->conv_shape_nr, 16 * h->conv_shape_nm,
                     (long)B * ((T + 16 * h->conv_shape_nm - 1) / (16 * h->conv_shape_nm)) * (2 * C / (16 * h->conv_shape_nr)));
      h->prof.note(PROF_WN_OUTPROJ, "convgemm16s_kernel<EpiResSkip16S<%d>, %d, %d> (v_mfma_f32_16x16x4_f32; %d x %d split-K workgroup tile, %ld workgroups)",
                   h->outp_shape_nm, h->outp_shape_nr, h->outp_shape_nm, 16 * h->outp_shape_nr, 16 * h->outp_shape_nm,
                   (long)B * ((T + 16 * h->outp_shape_nm - 1) / (16 * h->outp_shape_nm)) * (2 * C / (16 * h->outp_shape_nr)));
    }
  }
  for (int i = 0; i < L; ++i) {
    const int dil = l.dil[i];
    hipEvent_t ev0 = nullptr, ev1 = nullptr, eo0 = nullptr, eo1 = nullptr;   // fdx_prof_*: one of the two kernels, sampled
    h->prof.take(PROF_WN_CONVGATE, 2.0 * (2.0 * C) * (3.0 * C) * (double)B * T, ev0, ev1);
    h->prof.take(PROF_WN_OUTPROJ, 2.0 * (2.0 * C) * (double)C * (double)B * T, eo0, eo1);
    const float* Pl = (Pslab ? Pslab : h->P.f()) + kHalo + (size_t)i * 2 * C * ld;
    const long p_bs = (long)L * 2 * C * ld;
    const float* sbn = S + (size_t)(i + 1 < L ? i + 1 : 0) * C * ldn;
    const int skip_mode = (L == 1) ? 3 : (i == 0 ? 0 : (i + 1 == L ? 2 : 1));
    if (f16s) {               // fp16-split mode: both GEMMs as three v_mfma_f32_32x32x16_f16 per product block
      BfEpiGate eg{Pl, p_bs, ld, Zh, bsH, ld, C, 1.f / (kF16sWScale * kF16sYScale), kF16sZScale};
      if (f16s_small) {       // 64 x 64 tiles on v_mfma_f32_16x16x32_f16 (f16s64.hip.h)
        const uint4* WS = static_cast<const uint4*>(h->wn_f16s64.p) + (size_t)i * (f16s64_conv_groups(C) + f16s64_outp_groups(C));
        FDX_HIP(h, launch_f16s64(WS, reinterpret_cast<const uint4*>(Yh), bsH / 8, ld, C, dil, B, T, 2 * C, eg, s, ev0, ev1));
        BfEpiResSkip ers{X, SK, bsC, ld, A + l.outp[i].b_off, sbn, ldn, sb_bs, (i + 1 < L) ? Yh : nullptr, bsH, C, skip_mode, sqrtL,
                         (float)(1.0 / (double)sqrtL), 1.f / (kF16sWScale * kF16sZScale), kF16sYScale, keep, (long)ld};
        FDX_HIP(h, launch_f16s64(WS + f16s64_conv_groups(C), reinterpret_cast<const uint4*>(Zh), bsH / 8, ld, C, 0, B, T, 2 * C, ers, s, eo0, eo1));
        continue;
      }
      const uint4* WH = static_cast<const uint4*>(h->wn_f16s.p) + (size_t)i * (f16s_conv_groups(C) + f16s_outp_groups(C));
      FDX_HIP(h, launch_bf16lds<1>(WH, reinterpret_cast<const uint4*>(Yh), bsH / 8, ld, C, dil, B, T, 2 * C, eg, s, ev0, ev1));
      BfEpiResSkip er{X, SK, bsC, ld, A + l.outp[i].b_off, sbn, ldn, sb_bs, (i + 1 < L) ? Yh : nullptr, bsH, C, skip_mode, sqrtL,
                      (float)(1.0 / (double)sqrtL), 1.f / (kF16sWScale * kF16sZScale), kF16sYScale, keep, (long)ld};
      FDX_HIP(h, launch_bf16lds<1>(WH + f16s_conv_groups(C), reinterpret_cast<const uint4*>(Zh), bsH / 8, ld, C, 0, B, T, 2 * C, er, s, eo0, eo1));
      continue;
    }
    if (h->wn_arena_bf16) {   // bf16 storage mode: both GEMMs on v_mfma_f32_32x32x16_bf16
      WnBf16Layout bl;
      wn_bf16_layout(d, bl);
      const float4* WB = static_cast<const float4*>(h->wn_arena_bf16);
      __bf16* Yb = reinterpret_cast<__bf16*>(h->Yb.p) + (size_t)kHalo * 8;
      __bf16* Zb = reinterpret_cast<__bf16*>(h->Zb.p) + (size_t)kHalo * 8;
      const long bsB = (long)C * ld;               // bf16 elements per item
      if (h->wn_bf16_lds_ok && (long)B * ((T + 127) / 128) * (C / 64) >= bf16_lds_min_tiles()) {   // large column counts: LDS-tiled
        const uint4* WL = static_cast<const uint4*>(h->wn_bf16_lds.p);
        const uint4* Yg = reinterpret_cast<const uint4*>(Yb);
        const uint4* Zg = reinterpret_cast<const uint4*>(Zb);
        BfEpiGate eg{Pl, p_bs, ld, Zb, bsB, ld, C};
        FDX_HIP(h, launch_bf16lds(WL + bl.conv[i], Yg, bsB / 8, ld, C, dil, B, T, 2 * C, eg, s, ev0, ev1));
        BfEpiResSkip er{X, SK, bsC, ld, A + l.outp[i].b_off, sbn, ldn, sb_bs, (i + 1 < L) ? Yb : nullptr, bsB, C, skip_mode, sqrtL,
                        (float)(1.0 / (double)sqrtL)};
        FDX_HIP(h, launch_bf16lds(WL + bl.outp[i], Zg, bsB / 8, ld, C, 0, B, T, 2 * C, er, s, eo0, eo1));
        continue;
      }
      EpiGateB g{};
      g.out = Z; g.o_bs = bsC; g.ldo = ld; g.P = Pl; g.p_bs = p_bs; g.ldp = ld; g.C = C;
      g.outb = Zb; g.ob_bs = bsB;
      const ConvGeom gc{B, T, C / 16, 3, -dil, dil, bl.conv_mt};
      FDX_HIP(h, (launch_convgemm<2, true, PRE_NONE, EpiGateB, OPK_BF16>(gc, WB + bl.conv[i], reinterpret_cast<const float*>(Yb),
                                                                              bsB / 2, ld, 1.f, g, s, ev0, ev1)));
      EpiResSkipB r{};
      r.X = X; r.SK = SK; r.bs = bsC; r.ld = ld; r.bias = A + l.outp[i].b_off; r.C = C;
      r.Y = nullptr; r.sb = sbn; r.sb_ld = ldn; r.sb_bs = sb_bs;
      r.skip_mode = skip_mode; r.inv_div = sqrtL; r.r_inv_div = (float)(1.0 / (double)sqrtL);
      r.Yb = (i + 1 < L) ? Yb : nullptr; r.yb_bs = bsB;
      const ConvGeom go{B, T, C / 16, 1, 0, 0, bl.outp_mt};
      FDX_HIP(h, (launch_convgemm<2, true, PRE_NONE, EpiResSkipB, OPK_BF16>(go, WB + bl.outp[i], reinterpret_cast<const float*>(Zb),
                                                                                 bsB / 2, ld, 1.f, r, s, eo0, eo1)));
      continue;
    }
    {  // dilated conv k = 3 + gate (wavenet.py:107-115): 64 x 64 round-1 tile, or the shape picked for this geometry
      const ConvGeom g4{B, T, l.conv[i].cin8, 3, -dil, dil, l.conv[i].n_mtiles}, g2{B, T, l.conv[i].cin8, 3, -dil, dil, 2 * l.conv[i].n_mtiles};
      const int NRs = h->conv_shape_nr, NMs = h->conv_shape_nm;
      if (NRs == 4 && NMs == 4) {
        EpiGate16 g{Z, bsC, ld, Pl, p_bs, ld, C};
        FDX_HIP(h, launch_convgemm16(g4, reinterpret_cast<const float4*>(A + l.conv[i].w_off), Y, bsC, ld, g, s, ev0, ev1));
      } else {
        const void* W4 = A + l.conv[i].w_off;
        const void* W2 = h->wn_nr2.f() + h->wn_nr2_off[i];
        hipError_t e = hipErrorInvalidValue;
#define FDX_GATE_SHAPE(NR_, NM_)                                                                                              \
  if (NRs == NR_ && NMs == NM_) {                                                                                             \
    const EpiGate16S<NM_> gs{Z, bsC, ld, Pl, p_bs, ld, C};                                                                    \
    e = launch_convgemm16s<EpiGate16S<NM_>, NR_, NM_>(NR_ == 4 ? g4 : g2, NR_ == 4 ? W4 : W2, Y, bsC, ld, gs, s, ev0, ev1);   \
  }
        FDX_GATE_SHAPE(4, 5) FDX_GATE_SHAPE(4, 6) FDX_GATE_SHAPE(4, 7) FDX_GATE_SHAPE(4, 8)
        FDX_GATE_SHAPE(2, 4) FDX_GATE_SHAPE(2, 5) FDX_GATE_SHAPE(2, 6) FDX_GATE_SHAPE(2, 7) FDX_GATE_SHAPE(2, 8)
#undef FDX_GATE_SHAPE
        FDX_HIP(h, e);
      }
    }
    {  // out-projection + residual / skip (wavenet.py:117-120, 228): shape-adaptive 16x16x4 tiles (weights re-ordered at attach)
      const int NRo = h->outp_shape_nr, NMo = h->outp_shape_nm;
      const ConvGeom g4{B, T, l.outp[i].cin8, 1, 0, 0, l.outp[i].n_mtiles}, g2{B, T, l.outp[i].cin8, 1, 0, 0, 2 * l.outp[i].n_mtiles},
          g1{B, T, l.outp[i].cin8, 1, 0, 0, 4 * l.outp[i].n_mtiles};
      const void* W4 = h->wn_outp16.f() + h->wn_outp16_off4[i];
      const void* W2 = h->wn_outp16.f() + h->wn_outp16_off2[i];
      const void* W1 = h->wn_outp16.f() + h->wn_outp16_off1[i];
      hipError_t e = hipErrorInvalidValue;
#define FDX_OUTP_SHAPE(NR_, NM_)                                                                                                   \
  if (NRo == NR_ && NMo == NM_) {                                                                                                  \
    const EpiResSkip16S<NM_> rs{X, (i + 1 < L) ? Y : nullptr, SK, bsC, ld, A + l.outp[i].b_off, sbn, ldn, sb_bs, C, skip_mode, sqrtL, \
                                (float)(1.0 / (double)sqrtL), keep, (long)ld};                                                     \
    e = launch_convgemm16s<EpiResSkip16S<NM_>, NR_, NM_>(NR_ == 4 ? g4 : NR_ == 2 ? g2 : g1, NR_ == 4 ? W4 : NR_ == 2 ? W2 : W1, Z, bsC, ld, rs, s, eo0, eo1); \
  }
      FDX_OUTP_SHAPE(4, 4) FDX_OUTP_SHAPE(4, 5) FDX_OUTP_SHAPE(4, 6) FDX_OUTP_SHAPE(4, 7) FDX_OUTP_SHAPE(4, 8)
      FDX_OUTP_SHAPE(2, 4) FDX_OUTP_SHAPE(2, 5) FDX_OUTP_SHAPE(2, 6) FDX_OUTP_SHAPE(2, 7) FDX_OUTP_SHAPE(2, 8)
      FDX_OUTP_SHAPE(1, 4) FDX_OUTP_SHAPE(1, 5) FDX_OUTP_SHAPE(1, 6) FDX_OUTP_SHAPE(1, 7) FDX_OUTP_SHAPE(1, 8)
#undef FDX_OUTP_SHAPE
      FDX_HIP(h, e);
    }
  }
  {
    EpiBias e = epi_bias(H, bsC, ld, A + l.skip_proj.b_off, C, ACT_RELU);
    FDX_HIP(h, (run_gemm<true, false>(A, l.skip_proj, B, T, SK, bsC, ld, 0, 0, 1.f, e, s)));
  }
  if (fuse) {
    EpiUniPC e = *fuse;
    e.bias = A + l.out_proj.b_off; e.M = M; e.mask = mask; e.mask_ld = T;
    FDX_HIP(h, (run_gemm<true, false>(A, l.out_proj, B, T, H, bsC, ld, 0, 0, 1.f, e, s)));
  } else {
    EpiBias e = epi_bias(eps_out, o_bs, ldo, A + l.out_proj.b_off, M, ACT_NONE);
    e.mask = mask; e.mask_ld = T;
    e.tight = ldo != ld;   // fdx_wavenet_forward writes straight into the caller's [B][M][T] tensor
    FDX_HIP(h, (run_gemm<true, false>(A, l.out_proj, B, T, H, bsC, ld, 0, 0, 1.f, e, s)));
  }
  return FDX_OK;
}

extern "C" int fdx_wavenet_forward(fdx_handle h, const float* x, const float* t, int n_t, const uint8_t* x_mask,
                                   float* eps, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->wn_ok || !h->prepared) return fail(h, FDX_E_STATE, "fdx_wavenet_forward: call attach + prepare first");
  if (!x || !t || !eps) return fail(h, FDX_E_ARG, "fdx_wavenet_forward: null pointer");
  if (n_t != 1 && n_t != h->B) return fail(h, FDX_E_ARG, "diffusion_step must have 1 or B=%d entries, got %d", h->B, n_t);
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  const int M = h->wd.mel_channels, B = h->B, T = h->T, ld = h->ld;
  if (int rc = wn_embed(h, t, n_t, s)) return rc;
  hipLaunchKernelGGL(k_copy_rows, ew_grid(T, B * M), dim3(kEwBlock), 0, s, h->xin.f() + kHalo, (long)M * ld, ld, x,
                     (long)M * T, T, M, T, 1.f, (const uint8_t*)nullptr);
  return wn_forward_core(h, h->xin.f() + kHalo, 0, n_t == 1 ? 0 : 1, x_mask, eps, (long)M * T, T, s);
}

// ================================================================================================ sampler
static void launch_randn(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
  const size_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(k_randn, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, out, n, seed, offset);
}

extern "C" int fdx_randn(fdx_handle h, float* out, size_t n, uint64_t seed, uint64_t offset, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h || !out) return FDX_E_ARG;
  FDX_HIP(h, hipSetDevice(h->device));
  if (n) launch_randn(out, n, seed, offset, as_stream(st));
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

// The sampler body (every denoiser call + update rule of one run) as a stream of launches on `s`.  All buffers exist
// and every scalar is a kernel argument, so the same code is used to run eagerly and to record a hipGraph.
static int sampler_body(fdx_ctx* h, int kind, const float* tab, int n_rows, const float* step_noise, uint64_t seed,
                        const uint8_t* x_mask, hipStream_t s) {
  const int M = h->den_M, B = h->B, T = h->T, ld = h->ld;
  const long bs = (long)M * ld;
  const size_t bytes = (size_t)B * M * ld * sizeof(float);
  const dim3 grid = ew_grid(T, B * M), blk(kEwBlock);
  float* sx = h->sx.f() + kHalo;
  float* eps = h->EPS.f() + kHalo;
  // UniPC: the corrector (and the next predictor) ride in the denoiser's last projection's epilogue (EpiUniPC): one launch less per step
  static const bool fuse_unipc = [] { const char* e = getenv("FDX_UNIPC_FUSED"); return !e || atoi(e) != 0; }();
  auto model = [&](const float* xin, int col, bool masked, const EpiUniPC* fuse = nullptr) {
    // the one unmasked call of PLMS uses the conditioner slab of the UNMASKED conditioner (built in the set-up phase)
    // (exact-mask mode: the mask marks frames that do not exist, for an item run alone as well -- PLMS's unmasked call keeps it)
    const bool keep_mask = masked || h->ragged_keep;
    if (h->den_kind == 1) return fdx_cn_forward_core(h, xin, col, 0, keep_mask ? x_mask : nullptr, eps, bs, ld, s, !masked && h->cond_masked, fuse);
    if (h->den_kind == 2) return fdx_td_forward_core(h, xin, col, 0, keep_mask ? x_mask : nullptr, eps, bs, ld, s, !masked, fuse);
    const float* P = (!masked && h->cond_masked) ? h->P2.f() : nullptr;
    // (exact-mask mode: the mask marks frames that do not exist, for an item run alone as well -- PLMS's unmasked call keeps it)
    return wn_forward_core(h, xin, col, 0, (masked || h->ragged_keep) ? x_mask : nullptr, eps, bs, ld, s, P, fuse);
  };

  if (kind == FDX_SAMPLER_UNIPC) {
    float* xt = h->sxt.f() + kHalo; float* xb = h->sbase.f() + kHalo;
    float* m0 = h->sm[0].f() + kHalo; float* m1 = h->sm[1].f() + kHalo; float* mt = h->seps2.f() + kHalo;
    if (int rc = model(sx, 0, true)) return rc;
    hipLaunchKernelGGL(k_x0_pred, grid, blk, 0, s, m0, sx, eps, bs, ld, M, T, tab[1], tab[2]);
    bool pre_done = false;   // the predictor of step r was already computed by the fused post+pre launch of step r-1
    for (int r = 1; r < n_rows; ++r) {
      const float* row = tab + (size_t)r * FDX_ROW;
      const float sigma = row[1], alpha = row[2], c_x = row[3], c_m = row[4], aB = row[5], rk = row[6];
      const int order = (int)row[7], corr = (int)row[8];
      if (!pre_done)
        hipLaunchKernelGGL(k_unipc_pre, grid, blk, 0, s, xb, xt, sx, m0, m1, bs, ld, M, T, c_x, c_m, aB, rk, order);
      pre_done = false;
      if (corr && fuse_unipc) {   // (every denoiser since round 6: ConvNext's and the transformer's last projection take the same epilogue)
        EpiUniPC e{};
        e.x = sx; e.mt = mt; e.xbase = xb; e.xt = xt; e.m0 = m0; e.m1 = m1; e.bs = bs; e.ld = ld;
        e.sigma = sigma; e.alpha = alpha; e.aB = aB; e.rk = rk; e.rho0 = row[9]; e.rho1 = row[10]; e.order = order;
        if (r + 1 < n_rows) {
          const float* nx = row + FDX_ROW;
          e.n_cx = nx[3]; e.n_cm = nx[4]; e.n_aB = nx[5]; e.n_rk = nx[6]; e.n_order = (int)nx[7];
          pre_done = true;
        }
        if (int rc = model(xt, r, true, &e)) return rc;
        float* tmp = m1; m1 = m0; m0 = mt; mt = tmp;   // history shift (uni_pc.py:797-804)
      } else if (corr) {
        if (int rc = model(xt, r, true)) return rc;
        if (r + 1 < n_rows) {
          const float* nx = row + FDX_ROW;
          hipLaunchKernelGGL(k_unipc_post_pre, grid, blk, 0, s, sx, mt, xb, xt, eps, m0, m1, bs, ld, M, T, sigma, alpha, aB, rk,
                             order, row[9], row[10], nx[3], nx[4], nx[5], nx[6], (int)nx[7]);
          pre_done = true;
        } else {
          hipLaunchKernelGGL(k_unipc_post, grid, blk, 0, s, sx, mt, xb, xt, eps, m0, m1, bs, ld, M, T, sigma, alpha, aB, rk,
                             order, row[9], row[10]);
        }
        float* tmp = m1; m1 = m0; m0 = mt; mt = tmp;   // history shift (uni_pc.py:797-804)
      } else {
        FDX_HIP(h, hipMemcpyAsync(h->sx.p, h->sxt.p, bytes, hipMemcpyDeviceToDevice, s));
      }
    }
  } else if (kind == FDX_SAMPLER_NAIVE) {
    const size_t n_el = (size_t)B * M * T;
    for (int r = 0; r < n_rows; ++r) {
      const float* row = tab + (size_t)r * FDX_ROW;
      if (int rc = model(sx, r, true)) return rc;
      const float* nz = step_noise ? step_noise + (size_t)r * n_el : h->snoise.f();
      if (!step_noise) launch_randn(h->snoise.f(), n_el, seed, (uint64_t)r * ((n_el + 3) / 4), s);
      hipLaunchKernelGGL(k_naive_step, grid, blk, 0, s, sx, eps, nz, (long)M * T, T, bs, ld, M, T, row[1], row[2], row[3],
                         row[4], row[5], row[6], row[7]);
    }
  } else {  // PLMS
    float* xp = h->sxt.f() + kHalo; float* prime = h->seps2.f() + kHalo;
    // hist[0] = newest stored eps (noise_list[-1]) ... hist[2] = noise_list[-3]; `cur` takes this step's eps
    float* hist[3] = {h->shist[0].f() + kHalo, h->shist[1].f() + kHalo, h->shist[2].f() + kHalo};
    float* cur = h->shist[3].f() + kHalo;
    int n_hist = 0;
    for (int r = 0; r < n_rows; ++r) {
      const float* row = tab + (size_t)r * FDX_ROW;
      const float A = row[2], P = row[3], Q = row[4];
      if (int rc = model(sx, r, true)) return rc;
      FDX_HIP(h, hipMemcpyAsync(cur - kHalo, h->EPS.p, bytes, hipMemcpyDeviceToDevice, s));
      if (n_hist == 0) {
        hipLaunchKernelGGL(k_plms_pred, grid, blk, 0, s, xp, sx, cur, bs, ld, M, T, A, P, Q);
        if (int rc = model(xp, n_rows, false)) return rc;   // second call without masks (diffusion.py:285)
        hipLaunchKernelGGL(k_plms_blend, grid, blk, 0, s, prime, cur, eps, eps, eps, bs, ld, M, T, 0);
      } else {
        hipLaunchKernelGGL(k_plms_blend, grid, blk, 0, s, prime, cur, hist[0], hist[1], hist[2], bs, ld, M, T, n_hist);
      }
      hipLaunchKernelGGL(k_plms_pred, grid, blk, 0, s, sx, sx, prime, bs, ld, M, T, A, P, Q);
      float* freed = hist[2];
      hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = cur; cur = freed;
      if (n_hist < 3) ++n_hist;
    }
  }
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

static uint64_t fnv1a(const void* p, size_t n, uint64_t hsh = 1469598103934665603ULL) {
  const unsigned char* c = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < n; ++i) { hsh ^= c[i]; hsh *= 1099511628211ULL; }
  return hsh;
}

extern "C" int fdx_sampler_run(fdx_handle h, int kind, const float* tab, int n_rows, float* x, const float* step_noise,
                               uint64_t seed, const uint8_t* x_mask, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->prepared) return fail(h, FDX_E_STATE, "fdx_sampler_run: call attach + prepare first");
  if (!tab || n_rows <= 0 || !x) return fail(h, FDX_E_ARG, "fdx_sampler_run: bad table / x");
  if (kind != FDX_SAMPLER_NAIVE && kind != FDX_SAMPLER_UNIPC && kind != FDX_SAMPLER_PLMS)
    return fail(h, FDX_E_NOIMPL, "Unknown noise predictor: %d", kind);
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  const auto& d = h->wd;
  const int M = h->den_M, B = h->B, T = h->T, ld = h->ld;
  const long bs = (long)M * ld;
  const size_t bytes = (size_t)B * M * ld * sizeof(float);
  const dim3 grid = ew_grid(T, B * M), blk(kEwBlock);

  // ---------------------------------------------------------------- set-up (eager): buffers, step embeddings, x -> sx
  // timesteps of every model evaluation, embedded in one batch
  std::vector<float>& ts = h->ts_host;   // context-owned: outlives the (pageable, staged) async copy
  ts.clear();
  for (int r = 0; r < n_rows; ++r) ts.push_back(tab[(size_t)r * FDX_ROW]);
  if (kind == FDX_SAMPLER_PLMS) ts.push_back(tab[1]);   // t_prev of the first step (diffusion.py:285)
  FDX_HIP(h, h->tdev.ensure(ts.size() * 4, false, s));
  FDX_HIP(h, hipMemcpyAsync(h->tdev.p, ts.data(), ts.size() * 4, hipMemcpyHostToDevice, s));
  if (int rc = (h->den_kind == 1   ? fdx_cn_embed(h, h->tdev.f(), (int)ts.size(), s)
                : h->den_kind == 2 ? fdx_td_embed(h, h->tdev.f(), (int)ts.size(), s)
                                   : wn_embed(h, h->tdev.f(), (int)ts.size(), s)))
    return rc;

  FDX_HIP(h, h->sx.ensure(bytes, true, s));
  if (kind == FDX_SAMPLER_UNIPC) {
    FDX_HIP(h, h->sxt.ensure(bytes, true, s));
    FDX_HIP(h, h->sbase.ensure(bytes, true, s));
    for (auto& b : h->sm) FDX_HIP(h, b.ensure(bytes, true, s));
    FDX_HIP(h, h->seps2.ensure(bytes, true, s));
  } else if (kind == FDX_SAMPLER_NAIVE) {
    if (!step_noise) FDX_HIP(h, h->snoise.ensure((size_t)B * M * T * 4, false, s));
  } else {
    FDX_HIP(h, h->sxt.ensure(bytes, true, s));
    FDX_HIP(h, h->seps2.ensure(bytes, true, s));
    for (auto& b : h->shist) FDX_HIP(h, b.ensure(bytes, true, s));
    if (h->cond_masked && h->den_kind == 0) {   // conditioner slab of the unmasked conditioner for the one unmasked call
      FDX_HIP(h, h->P2.ensure((size_t)B * d.residual_layers * 2 * d.residual_channels * ld * sizeof(float), false, s));
      if (int rc = wn_cond_slab(h, h->condraw.f(), h->P2.f(), s)) return rc;
    }
    if (h->cond_masked && h->den_kind == 1)
      if (int rc = fdx_cn_plms_setup(h, s)) return rc;
  }
  hipLaunchKernelGGL(k_copy_rows, grid, blk, 0, s, h->sx.f() + kHalo, bs, ld, x, (long)M * T, T, M, T, 1.f, (const uint8_t*)nullptr);
  if (x_mask) {   // private copy: a stable address for the recorded graph, whatever tensor the caller passes next time
    FDX_HIP(h, h->maskbuf.ensure((size_t)B * T, false, s));
    FDX_HIP(h, hipMemcpyAsync(h->maskbuf.p, x_mask, (size_t)B * T, hipMemcpyDeviceToDevice, s));
    x_mask = static_cast<const uint8_t*>(h->maskbuf.p);
  }

  // ---------------------------------------------------------------- body: eager, or one hipGraph launch
  // ~42 launches per step x 100 steps cost the host ~45 ms per run when issued one by one (half of the GPU time at
  // 10 s / batch 1, more than the GPU time for short utterances).  The body only depends on (sampler, table, geometry,
  // buffer addresses), so it is recorded once per such key on a private capture stream and replayed with one
  // hipGraphLaunch on the caller's stream.  Not graphed: profiling runs (per-dispatch events) and the DDPM sampler (its
  // per-step noise is either a caller tensor whose address changes or a Philox seed that is a kernel argument).
  const bool graphable = h->use_graphs && !h->prof.on && kind != FDX_SAMPLER_NAIVE;
  if (!graphable) {
    if (int rc = sampler_body(h, kind, tab, n_rows, step_noise, seed, x_mask, s)) return rc;
  } else {
    const uint64_t tab_hash = fnv1a(tab, (size_t)n_rows * FDX_ROW * sizeof(float));
    auto make_key = [&]() {
      const uint64_t parts[] = {(uint64_t)kind, (uint64_t)n_rows, (uint64_t)B, (uint64_t)T, (uint64_t)(x_mask != nullptr),
                                (uint64_t)(uintptr_t)h->wn_arena, (uint64_t)h->cond_masked, h->alloc_gen,
                                (uint64_t)h->den_kind, (uint64_t)(uintptr_t)h->cn, (uint64_t)(uintptr_t)h->td,
                                (uint64_t)(uintptr_t)h->ragged_keep, h->items_hash};
      return fnv1a(parts, sizeof parts, tab_hash);
    };
    uint64_t key = make_key();
    fdx_ctx::GraphEntry* hit = nullptr;
    for (auto& g : h->graphs) if (g.key == key) hit = &g;
    if (!hit) {
      if (!h->cap_stream) FDX_HIP(h, hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
      hipGraph_t graph = nullptr;
      FDX_HIP(h, hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
      const int rc = sampler_body(h, kind, tab, n_rows, step_noise, seed, x_mask, h->cap_stream);
      const hipError_t ec = hipStreamEndCapture(h->cap_stream, &graph);
      if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      FDX_HIP(h, ec);
      hipGraphExec_t exec = nullptr;
      FDX_HIP(h, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      // Workspaces first touched INSIDE the body (a handle's very first run) were allocated while recording: the recording holds their final
      // addresses, so it is filed under the allocation generation AFTER the capture -- round 5 filed it under the one before, and the second
      // run of every handle recorded the same 4 k (transformer: 13 k) nodes once more (bench.py `first_call_ms.sampler_graphs`: 2 -> 1).
      key = make_key();
      // LRU cache.  Keys change with geometry / schedule / reallocation; ragged serving (pipeline.synthesize pads every micro-batch
      // to a 64-frame bucket) cycles through (batch size, bucket) pairs -- a few dozen for 6-10 s utterances in batches of <= 8 --
      // so the cache holds that many (an instantiated sampler graph is ~4 k kernel nodes of host memory, no device memory).
      if (h->graphs.size() >= (size_t)h->graph_cap) {
        size_t lru = 0;
        for (size_t i = 1; i < h->graphs.size(); ++i) if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
        (void)hipGraphExecDestroy(h->graphs[lru].exec);
        h->graphs.erase(h->graphs.begin() + lru);
      }
      h->graphs.push_back({key, exec, 0});
      hit = &h->graphs.back();
      ++h->graph_captures;
    }
    hit->last_use = ++h->graph_clock;
    ++h->graph_launches;
    FDX_HIP(h, hipGraphLaunch(hit->exec, s));
  }

  // ---------------------------------------------------------------- finish (eager): sx -> x
  hipLaunchKernelGGL(k_copy_rows, grid, blk, 0, s, x, (long)M * T, T, h->sx.f() + kHalo, bs, ld, M, T, 1.f, (const uint8_t*)nullptr);
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

// keep[b][kHalo + t] = mask[b][t] ? 0 : 1 (padded rows; the pads stay 0 from the buffer's zero fill)
static __global__ void k_keep_from_mask(float* __restrict__ keep, int ld, const uint8_t* __restrict__ mask, int B, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t < T) keep[(long)b * ld + kHalo + t] = mask[(long)b * T + t] ? 0.f : 1.f;
}

extern "C" int fdx_sampler_run_ragged(fdx_handle h, int kind, const float* tab, int n_rows, float* x, const float* step_noise,
                                      uint64_t seed, const uint8_t* x_mask, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->prepared) return fail(h, FDX_E_STATE, "fdx_sampler_run_ragged: call attach + prepare first");
  if (!x_mask) return fail(h, FDX_E_ARG, "fdx_sampler_run_ragged: null mask");
  const int B = h->B, T = h->T, ld = h->ld;
  if (h->den_kind != 0 && h->n_items() && (B != 1 || h->items_T != T))
    return fail(h, FDX_E_STATE, "fdx_sampler_run_ragged: the item layout (fdx_sampler_set_items) describes a row of %d frames, the prepared batch is %d x %d",
                h->items_T, B, T);
  // Attention does not stop at holes and positions count per item: without the layout the result would silently not be the per-item one
  // fishdx.h promises.  (prepare cannot know a ragged run follows, so the check lives here.)
  if ((h->den_kind == 2 || (h->den_kind == 1 && fdx_cn_has_attention(h))) && h->n_items() == 0)
    return fail(h, FDX_E_STATE, "fdx_sampler_run_ragged: the prepared denoiser has attention layers: describe the row's items with fdx_sampler_set_items (before prepare) first");
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  FDX_HIP(h, h->keepbuf.ensure((size_t)B * ld * sizeof(float), true, s));
  hipLaunchKernelGGL(k_keep_from_mask, dim3((T + 255) / 256, B), dim3(256), 0, s, h->keepbuf.f(), ld, x_mask, B, T);
  // the conv's input must read 0 wherever no frame exists: what earlier runs (other masks) left there is cleared
  if (h->den_kind == 0) FDX_HIP(h, hipMemsetAsync(h->Y.p, 0, (size_t)B * h->wd.residual_channels * ld * sizeof(float), s));
  // (ConvNext / transformer: every per-frame op is column-local and the depthwise conv masks its own input, convnext.hip k_dwconv_stats; the
  // attention layers need the item layout -- refused above without it)
  h->ragged_keep = h->keepbuf.f() + kHalo;
  const int rc = fdx_sampler_run(h, kind, tab, n_rows, x, step_noise, seed, x_mask, st);
  h->ragged_keep = nullptr;
  return rc;
}

struct ItemChunk { static constexpr int kN = 32; int base, n; int v[2 * kN]; };
static __global__ void k_items_fill(int4* __restrict__ items, ItemChunk c) {
  const int i = threadIdx.x;
  if (i < c.n) items[c.base + i] = make_int4(c.v[2 * i], c.v[2 * i + 1], 0, 0);
}

// pidx[t] = t - offset of the item frame t belongs to (0 in holes)
static __global__ void k_items_pidx(int* __restrict__ pidx, const int4* __restrict__ items, int n_items, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  int p = 0;
  for (int i = 0; i < n_items; ++i) {
    const int4 it = items[i];
    if (t >= it.x && t < it.x + it.y) p = t - it.x;
  }
  pidx[t] = p;
}

extern "C" int fdx_sampler_set_items(fdx_handle h, const int* offsets, const int* lens, int n_items, int T, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (n_items < 0 || (n_items > 0 && (!offsets || !lens || T <= 0))) return fail(h, FDX_E_ARG, "fdx_sampler_set_items: bad arguments");
  h->prepared = false;                      // the hoisted condition path depends on the layout (positions): prepare again
  h->items.clear();
  h->items_hash = 0; h->items_max_len = 0; h->items_T = 0;
  if (n_items == 0) return FDX_OK;
  // validated into locals and committed only when every item passed: a failure at item i > 0 must not leave half a layout behind
  int end = 0, max_len = 0;
  std::vector<int> items_new, packed((size_t)n_items * 4, 0);
  for (int i = 0; i < n_items; ++i) {
    if (offsets[i] % 32 || offsets[i] < end || lens[i] <= 0 || offsets[i] + lens[i] > T)
      return fail(h, FDX_E_ARG, "fdx_sampler_set_items: item %d = [%d, %d) must start at a multiple of 32, follow item %d and end inside the row of %d frames",
                  i, offsets[i], offsets[i] + lens[i], i - 1, T);
    end = offsets[i] + lens[i];
    items_new.push_back(offsets[i]); items_new.push_back(lens[i]);
    packed[4 * i] = offsets[i]; packed[4 * i + 1] = lens[i];
    max_len = std::max(max_len, lens[i]);
  }
  h->items.swap(items_new);
  h->items_max_len = max_len;
  h->items_T = T;
  // what a recorded sampler graph depends on: the launches' grids (number of items, query blocks of the longest, finest key split) -- NOT where
  // the items lie or how long each is: the kernels read offsets / lengths from the device table, rewritten above before any replay.  (Keyed by
  // the full layout, every micro-batch of a serving stream would have re-captured its 15 k-node graph.)
  int ks_max = 1;
  for (int i = 0; i < n_items; ++i) ks_max = std::max(ks_max, attn_ksplit_of(1, lens[i], lens[i], 0));
  const int key_parts[4] = {n_items, (h->items_max_len + 63) / 64, ks_max, T};   // (64-query blocks: the combine kernel's grid; the attention's 128-query blocks follow)
  h->items_hash = fnv1a(key_parts, sizeof key_parts, 0x9e3779b97f4a7c15ull);
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  FDX_HIP(h, h->items_dev.ensure(packed.size() * sizeof(int), false, s));
  FDX_HIP(h, h->pidx_dev.ensure((size_t)T * sizeof(int), false, s));
  // The table travels as KERNEL ARGUMENTS (copied at launch, 32 items per launch): no host buffer whose lifetime an asynchronous copy
  // would depend on, and no synchronisation in a call the serving loop makes per micro-batch.
  for (int i0 = 0; i0 < n_items; i0 += ItemChunk::kN) {
    ItemChunk c{};
    c.base = i0; c.n = std::min(ItemChunk::kN, n_items - i0);
    for (int i = 0; i < c.n; ++i) { c.v[2 * i] = packed[4 * (i0 + i)]; c.v[2 * i + 1] = packed[4 * (i0 + i) + 1]; }
    hipLaunchKernelGGL(k_items_fill, dim3(1), dim3(64), 0, s, static_cast<int4*>(h->items_dev.p), c);
  }
  hipLaunchKernelGGL(k_items_pidx, dim3((T + 255) / 256), dim3(256), 0, s, static_cast<int*>(h->pidx_dev.p), static_cast<const int4*>(h->items_dev.p), n_items, T);
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

extern "C" int fdx_graph_stats(fdx_handle h, long* captures, long* launches, int* cached) {
  if (!h) return FDX_E_ARG;
  if (captures) *captures = h->graph_captures;
  if (launches) *launches = h->graph_launches;
  if (cached) *cached = (int)h->graphs.size();
  return FDX_OK;
}

extern "C" int fdx_q_sample(fdx_handle h, const float* src, int B, int M, int T, int normalise, const float* spec_min,
                            const float* spec_max, int n_spec, float sqrt_ac, float sqrt_1m_ac, const float* noise, float* out,
                            fdx_stream st) {
  GenScope gen_scope(h);
  if (!h || !src || !out || B <= 0 || M <= 0 || T <= 0) return FDX_E_ARG;
  if (normalise && (!spec_min || !spec_max || (n_spec != 1 && n_spec != T)))
    return fail(h, FDX_E_ARG, "fdx_q_sample: spec_min / spec_max must have 1 or T=%d entries (they broadcast over the last axis), got %d", T, n_spec);
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  const float* dmin = nullptr; const float* dmax = nullptr;
  if (normalise) {
    FDX_HIP(h, h->scratch_a.ensure(2 * (size_t)n_spec * 4, false, s));
    FDX_HIP(h, hipMemcpyAsync(h->scratch_a.p, spec_min, n_spec * 4, hipMemcpyHostToDevice, s));
    FDX_HIP(h, hipMemcpyAsync(h->scratch_a.f() + n_spec, spec_max, n_spec * 4, hipMemcpyHostToDevice, s));
    dmin = h->scratch_a.f(); dmax = h->scratch_a.f() + n_spec;
  }
  const size_t n = (size_t)B * M * T;
  hipLaunchKernelGGL(k_q_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, src, noise, n, T, normalise, dmin, dmax, n_spec,
                     noise != nullptr, sqrt_ac, sqrt_1m_ac);
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

extern "C" int fdx_denorm_spec(fdx_handle h, const float* x, int B, int M, int T, const float* spec_min,
                               const float* spec_max, int n_spec, float* mel, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h || !x || !mel || !spec_min || !spec_max) return FDX_E_ARG;
  if (n_spec != 1 && n_spec != M) return fail(h, FDX_E_ARG, "spec_min and spec_max must be either of length 1 or mel_channels");
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  FDX_HIP(h, h->scratch_a.ensure(2 * (size_t)n_spec * 4, false, s));
  FDX_HIP(h, hipMemcpyAsync(h->scratch_a.p, spec_min, n_spec * 4, hipMemcpyHostToDevice, s));
  FDX_HIP(h, hipMemcpyAsync(h->scratch_a.f() + n_spec, spec_max, n_spec * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_denorm_transpose, dim3((T + 31) / 32, (M + 31) / 32, B), dim3(256), 0, s, mel, x, M, T,
                     h->scratch_a.f(), h->scratch_a.f() + n_spec, n_spec);
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}
