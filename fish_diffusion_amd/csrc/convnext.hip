// convnext.hip -- the ConvNext denoiser: fish_diffusion/modules/convnext.py:12-92 (ConvNeXtBlock),
// :95-152 (CrossAttentionBlock), :155-262 (ConvNext, with and without cross_attention), registered as DENOISERS "ConvNextDenoiser"
// (archs/diffsinger/diffusions/builder.py:12).  Same call contract as the WaveNet denoiser, so the sampler loop of
// wavenet.hip drives it through the two hooks fdx_cn_embed / fdx_cn_forward_core.
//
// Per denoiser call (B items, T frames, D = dim, H = D * mlp_factor, L layers):
//   1 x  input_projection      GEMM [D x M]     epilogue: +bias, GELU, mask                                   :231-239
//   L x  dwconv + LN stats     VALU             u = dwconv_d(mask(x + step_l + cond_l)); per-frame channel statistics of u  :66-79
//   L x  pwconv1 (LayerNorm)   GEMM [H x D]     LayerNorm folded in (PRE_LN: group-offset term, * rstd); epilogue: +bias, GELU   :80-82
//   L x  pwconv2               GEMM [D x H]     epilogue: x = x + gamma * (. + bias), mask                    :83-92
//   1 x  output_projection.0   GEMM [D x D]     epilogue: +bias, GELU                                         :256
//   1 x  output_projection.2   GEMM [M x D]     epilogue: +bias, mask                                         :256-258
// Hoisted, exactly as for the WaveNet: the conditioner path (conditioner_projection MLP, then the L condition_projection
// 1x1 convs as ONE [L*D x D] GEMM) runs once per utterance batch (fdx_convnext_prepare); the step-embedding MLP and the L
// diffusion_step_projection 1x1 convs run once per sampler run for all timesteps.
//
// cross_attention = True (convnext.py:186-193,246-250): a CrossAttentionBlock -- one nn.TransformerDecoderLayer (declayer.hip.h)
// on x + diffusion_step_projection(step) + pos * scale_q, attending to condition + pos * scale_k -- in front of every
// `cross_every_n_layers`-th ConvNeXt block, and the ConvNeXt blocks then run WITHOUT the condition term.  The memory of every
// cross block is step-invariant, so its key / value projection is hoisted into fdx_convnext_prepare as well.
#include "common.hip.h"
#include "convplan.hip.h"
#include "gemmplan.hip.h"
#include "elementwise.hip.h"
#include "declayer.hip.h"
#include "convgemm16s.hip.h"

#include <cmath>

using namespace fdx;

namespace {

struct CnLayout {
  PackedW in_proj, emb1, emb3, cond0, cond2, dsp, cproj, out0, out2;   // dsp / cproj: all layers concatenated along rows
  std::vector<PackedW> pw1, pw2, pw2w;   // pw2: 32-row tiles (small grids), pw2w: the same weights in 64-row tiles (large grids)
  std::vector<size_t> dw_w, dw_b, gamma, lnR, lnRs;   // lnRs: [round_up(H, 64)] whole-row sums (PRE_LNP)
  //   // (norm.weight / norm.bias live folded inside pw1; lnR: [round_up(H, 64)][16] group row sums)
  std::vector<int> dil;
  // cross-attention variant: one entry per CrossAttentionBlock, in front of ConvNeXt block `at_layer`
  struct Cross { TdLayer dec; size_t scale_q, scale_k; int at_layer; };
  std::vector<Cross> cross;
  size_t pos = 0;            // positional_embedding [kCnPositions][D] (one copy: the blocks' buffers are identical, checked at pack time)
  size_t total_floats = 0;
};
constexpr int kCnPositions = 4096;   // CrossAttentionBlock.get_embedding(num_embeddings=4096), convnext.py:114
constexpr int kCrossTensors = 3 + 18 + 2;

// pwconv1's LayerNorm fold and tile height (round 6, tools/cnbench.py at T = 861, us per denoiser call, batch 1 | batch 8):
//   round 5: depthwise-conv output stored group-centred, PRE_LN's 16-term group correction, 64-row tiles      1043 | 6119
//   stored as it is, PRE_LNP (rstd (acc - mean rowsum): one multiply-subtract per output), 64-row tiles        1032
//   PRE_LN with 32-row tiles (twice the workgroups, each paying the correction)                                 1089
//   PRE_LNP with 32-row tiles (896 workgroups, four co-resident per CU instead of 448 at 1.75 per CU)            1001 | 5783   <- default
// A/B switches (decide the arena layout / the operand form: read once per process): FDX_CN_PW1_RB=2 (64-row tiles), FDX_CN_LNP=0 (centred form).
int cn_pw1_rb() { static const int v = [] { const char* e = getenv("FDX_CN_PW1_RB"); const int k = e ? atoi(e) : 0; return k == 2 ? 2 : 1; }(); return v; }
// pwconv1 on the shape-adaptive split-K 16x16x4 family of the residual-block GEMMs (convgemm16s.hip.h, PRE_LNP + bias + GELU): at batch 1 x 861
// frames 256 workgroups of 64 x 112 (one per CU) instead of 896 of 32 x 64 -- 24.3 -> ~19.6 us per launch; FDX_CN_PW1_16S=0 keeps the 32x32x2 kernel.
// Taken for EVERY geometry when on (an item's result must not depend on the batch it rode in).  Needs the PRE_LNP form and 32-row packed weights.
bool cn_pw1_16s() { static const bool v = [] { const char* e = getenv("FDX_CN_PW1_16S"); return !e || atoi(e) != 0; }(); return v; }
bool cn_lnp() { static const bool v = [] { const char* e = getenv("FDX_CN_LNP"); return !e || atoi(e) != 0; }(); return v; }
int cn_n_cross(const fdx_convnext_desc& d) { return d.cross_attention > 0 ? (d.num_layers + d.cross_attention - 1) / d.cross_attention : 0; }

int cn_validate(const fdx_convnext_desc* d) {
  if (!d) return fail(nullptr, FDX_E_ARG, "null convnext desc");
  if (d->dim < 32 || d->dim % 32 || d->dim > 512) return fail(nullptr, FDX_E_ARG, "convnext: dim must be a multiple of 32 in [32, 512], got %d", d->dim);
  if (d->mlp_factor < 1 || d->mlp_factor > 8) return fail(nullptr, FDX_E_ARG, "convnext: mlp_factor out of range");
  if (d->mel_channels <= 0 || d->mel_channels % 8 || d->condition_dim <= 0 || d->condition_dim % 8)
    return fail(nullptr, FDX_E_ARG, "convnext: mel_channels and condition_dim must be multiples of 8");
  if (d->num_layers <= 0) return fail(nullptr, FDX_E_ARG, "convnext: num_layers must be positive");
  if (d->dilation_cycle < 1 || d->dilation_cycle > 4) return fail(nullptr, FDX_E_ARG, "convnext: dilation_cycle %d unsupported (3 * dilation must fit the %d-column halo)", d->dilation_cycle, kHalo);
  if (d->cross_attention < 0) return fail(nullptr, FDX_E_ARG, "convnext: cross_attention (= cross_every_n_layers, 0 = off) must be >= 0");
  if (d->cross_attention > 0 && d->dim != 128 && d->dim != 256 && d->dim != 512)
    return fail(nullptr, FDX_E_ARG, "convnext: cross-attention needs dim 128, 256 or 512 (8 heads of 16 / 32 / 64), got %d", d->dim);
  return FDX_OK;
}

void cn_layout(const fdx_convnext_desc& d, CnLayout& l) {
  const int D = d.dim, H = d.dim * d.mlp_factor, L = d.num_layers;
  size_t cur = 0;
  l.in_proj = plan32(cur, D, d.mel_channels);   // D- and M-row GEMMs: 32-row tiles (twice the workgroups)
  l.emb1 = plan64(cur, H, D);
  l.emb3 = plan64(cur, D, H);
  l.cond0 = plan64(cur, H, d.condition_dim);
  l.cond2 = plan64(cur, D, H);
  const int NC = cn_n_cross(d);
  l.dsp = plan64(cur, (L + NC) * D, D);      // rows [L*D, (L+NC)*D): the cross blocks' own diffusion_step_projection
  l.cproj = plan64(cur, L * D, D);
  l.pw1.clear(); l.pw2.clear(); l.pw2w.clear(); l.dw_w.clear(); l.dw_b.clear(); l.gamma.clear(); l.lnR.clear(); l.lnRs.clear(); l.dil.clear();
  for (int i = 0; i < L; ++i) {
    l.pw1.push_back(cn_pw1_rb() == 1 ? plan32(cur, H, D) : plan64(cur, H, D));
    l.pw2.push_back(plan32(cur, D, H));
    l.pw2w.push_back(plan64(cur, D, H));
    l.dw_w.push_back(cur); cur += round_up(D * 7, 64);
    for (auto* v : {&l.dw_b, &l.gamma}) { v->push_back(cur); cur += round_up(D, 64); }
    l.lnR.push_back(cur); cur += (size_t)round_up(H, 64) * 16;
    l.lnRs.push_back(cur); cur += (size_t)round_up(H, 64);
    l.dil.push_back(1 << (i % d.dilation_cycle));
  }
  l.cross.clear();
  l.pos = cur;
  if (NC) cur += (size_t)round_up(kCnPositions * D, 64);
  for (int c = 0; c < NC; ++c) {
    CnLayout::Cross x{};
    x.at_layer = c * d.cross_attention;
    plan_declayer(cur, x.dec, D, H);
    x.scale_q = cur; cur += 64;
    x.scale_k = cur; cur += 64;
    l.cross.push_back(x);
  }
  l.out0 = plan32(cur, D, D);
  l.out2 = plan32(cur, d.mel_channels, D);
  l.total_floats = cur;
}

// X[b][c][t] += (sb ? sb[c] : 0) + pos[t][c] * scale    (CrossAttentionBlock.forward: x + diffusion_step_projection(step), then
// + positional_embedding * position_scale_query, convnext.py:127-136; with sb == null and out != in: condition + pos * scale_key)
__global__ void k_cn_addpos(float* out, const float* in, long bs, int ld, const float* __restrict__ sb, int sb_ld,
                            int sb_bs, const float* __restrict__ pos, const float* __restrict__ scale, int D, int T,
                            const int* __restrict__ pidx /* exact-ragged rows: position inside the item; null = the column */) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const int b = blockIdx.y / D, c = blockIdx.y - b * D;
  const long o = b * bs + (long)c * ld + t;
  float v = in[o];
  if (sb) v = v + sb[(long)c * sb_ld + b * sb_bs];
  out[o] = v + pos[(long)(pidx ? pidx[t] : t) * D + c] * scale[0];
}

// ------------------------------------------------------------------------------------------------ dwconv + LayerNorm
// k_dwconv_stats, (T/64) x (D/32) x B workgroups (the op is latency-, not bandwidth-bound: it wants many small workgroups; one
// workgroup owning all D channels of a frame tile ran 33.7 us per layer at T = 861):
//   v = mask(x + step + cond) for 32 channels x (64 + 6*dil) frames staged ONCE into LDS (zero outside [0, T): exactly the zero
//   padding of the reference's depthwise Conv1d applied after the adds and the mask, convnext.py:64-77); u = dwconv(v) + bias
//   from LDS -> U; per frame the group's mean and centred sum of squares (two-pass, over its 32 channels) -> ST[b][t][{mean,M2}][16].
// The LayerNorm itself is folded into the pwconv1 GEMM (convgemm.hip.h PRE_LN): U holds every channel centred by the mean of its
// OWN group of 32 channels (exact); the GEMM's epilogue combines the D/32 group statistics of the lane's two frames (Chan et
// al.), adds the group-offset term sum_g R[row][g] (mean_g - mean) (R = per-group row sums of the folded weights) and scales
// by rstd[t]; the affine part (norm.weight, norm.bias) is folded into pwconv1's weights and bias at pack time:
//   W1 (((u - mean) rstd) w + b) + b1  =  rstd (W1 diag(w)) (u - mean) + (W1 b + b1).
constexpr int kCnCh = 32;            // channels per workgroup
constexpr int kCnMaxW = 64 + 6 * 8;  // widest staged row (dilation 8)

// (argument order: what the first loads need -- X, geometry, CP, SB -- fills the 14 dwords the dispatcher preloads into SGPRs
// (-amdgpu-kernarg-preload-count=14, _build.py); the rest is fetched from the kernarg segment while those loads are in flight.  With the
// outputs first, every group of arguments was a kernarg round trip in front of its first use.)
__global__ __launch_bounds__(256) void k_dwconv_stats(const float* __restrict__ X, long bs, int ld, int T, int dil, int D,
                                                      const float* __restrict__ CP, long cp_bs,            // layer slab [D][ld]
                                                      const float* __restrict__ SB, int sb_ld, int sb_bs,  // [D][sb_ld], column = step
                                                      const float* __restrict__ dw_w, const float* __restrict__ dw_b,
                                                      const uint8_t* __restrict__ mask, float* __restrict__ U, float* __restrict__ ST, int centre) {
  __shared__ float v[kCnCh][kCnMaxW];
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t0 = blockIdx.x * 64, g = blockIdx.y, b = blockIdx.z;
  const int W = 64 + 6 * dil, tl = t0 - 3 * dil;
  // ---- stage v: wave wv loads channels wv, wv + 4, ...; lanes cover the W columns in two chunks.  Order of issue: (1) the frame's x and
  // condition values, which need only preloaded arguments; (2) ONE kernarg fetch for everything else (touched together below: left to their first
  // uses they were a round trip each); (3) the mask bytes, step projections and taps, all unconditional; then the first wait.
  bool inside[2];
  int col[2], tq[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    col[q] = lane + 64 * q;
    const int tt = tl + col[q];
    tq[q] = min(max(tt, 0), T - 1);   // loads are unconditional (all in flight at once) from a clamped frame, then selected
    inside[q] = col[q] < W && tt >= 0 && tt < T;
  }
  const float* xb = X + b * bs;
  const float* cb = CP + b * cp_bs;
  float xv[kCnCh / 4][2], cv[kCnCh / 4][2], sbv[kCnCh / 4];
#pragma unroll
  for (int j = 0; j < kCnCh / 4; ++j) {
    const int c = g * kCnCh + wv + 4 * j;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      xv[j][q] = xb[(long)c * ld + tq[q]];
      cv[j][q] = CP ? cb[(long)c * ld + tq[q]] : 0.f;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if ((reinterpret_cast<uintptr_t>(dw_w) | reinterpret_cast<uintptr_t>(dw_b) | reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(ST) |
       reinterpret_cast<uintptr_t>(mask) | (uintptr_t)(unsigned)sb_ld | (uintptr_t)(unsigned)sb_bs) == 0) return;
  const uint8_t* mrow = mask ? mask + (long)b * T : reinterpret_cast<const uint8_t*>(dw_w);   // no mask: any readable bytes, ignored
  uint8_t mb[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) mb[q] = mrow[mask ? tq[q] : 0];      // (used below, behind the other loads' issue: a use here is a vmcnt(0))
  // (the wave's 8 step-projection values as one vector load, lane j & 7 -> channel wv + 4 j: as wave-uniform loads they were 8 scalar loads
  // with a wait behind each, in front of everything else)
  const float sbl = SB[(long)(g * kCnCh + wv + 4 * (lane & 7)) * sb_ld + b * sb_bs];
  // the 8 x 7 taps + biases this wave's channels need, fetched with the frame (behind the barrier they would be a second memory round trip)
  // (ONE vector load per wave: its 8 channels' 56 taps are contiguous, lanes 56..63 take the 8 biases; as 64 wave-uniform loads hipcc issued
  // 64 scalar loads with a wait behind each)
  const int c8 = g * kCnCh + wv * 8;
  const float wl = lane < 56 ? dw_w[c8 * 7 + lane] : dw_b[c8 + lane - 56];
  __builtin_amdgcn_sched_barrier(0);
  bool ok[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) ok[q] = inside[q] && !(mask && mb[q]);
#pragma unroll
  for (int j = 0; j < kCnCh / 4; ++j) sbv[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sbl), j));
#pragma unroll
  for (int j = 0; j < kCnCh / 4; ++j)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (col[q] < W) v[wv + 4 * j][col[q]] = ok[q] ? (CP ? (xv[j][q] + sbv[j]) + cv[j][q] : xv[j][q] + sbv[j]) : 0.f;   // CP == null: no condition term
  __syncthreads();
  // ---- conv from LDS: thread (lane = frame, wave = 8 channels)
  const int t = t0 + lane;
  float u[8];
  float s1 = 0.f;
  auto bcast = [&](int src) __attribute__((always_inline)) {   // wave-uniform value of lane `src` (a scalar register)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), src));
  };
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = wv * 8 + j;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc += bcast(j * 7 + k) * v[ch][lane + k * dil];
    u[j] = acc + bcast(56 + j);
    s1 += u[j];
  }
  red[wv][lane] = s1;
  __syncthreads();
  const float mean_g = (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) * (1.f / kCnCh);
  __syncthreads();
  float s2 = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float dlt = u[j] - mean_g; s2 += dlt * dlt; }
  red[wv][lane] = s2;
  __syncthreads();
  if (t >= T) return;
  if (wv == 0) {
    float* st = ST + ((long)b * T + t) * 32 + g;
    st[0] = mean_g;
    st[16] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  }
  float* ub = U + b * bs + t;
  const float sub = centre ? mean_g : 0.f;       // group-centred (PRE_LN) or as it is (PRE_LNP)
#pragma unroll
  for (int j = 0; j < 8; ++j) ub[(long)(g * kCnCh + wv * 8 + j) * ld] = u[j] - sub;
}

struct CnBufs {
  DevBuf X, N, G, H2, condp, c1, c2, CP, ST;
  DevBuf c2raw, CP2;   // PLMS + cond_masks: condition / per-layer projections of the UNMASKED conditioner (diffusion.py:285)
  DevBuf E, Hm, S0, SB;
  DevBuf QKV, O, MEM, KVc, KVc2, cmask, AP, AML;   // cross-attention variant: decoder-layer scratch; hoisted keys / values per cross block ([NC][2D] rows)
  int ldn = 0, n_emb = 0;
};

}  // namespace

struct fdx_cn_state {
  bool ok = false;
  fdx_convnext_desc d{};
  CnLayout l;
  const float* arena = nullptr;
  CnBufs b;
  // pwconv1's weights once more in the 16x16x4 fragment orders (NR = 4 and NR = 2), derived on the device at attach (cn_pw1_16s)
  DevBuf pw1_16;
  std::vector<size_t> pw1_off4, pw1_off2;
  bool pw1_16_ok = false;
  int pw1_nr = 4, pw1_nm = 4;     // tile shape for the prepared geometry
};

static fdx_cn_state* cn(fdx_ctx* h) {
  if (!h->cn) h->cn = new fdx_cn_state();
  return static_cast<fdx_cn_state*>(h->cn);
}
void fdx_cn_free(void* p) { delete static_cast<fdx_cn_state*>(p); }
bool fdx_cn_has_attention(fdx_ctx* h) { return h->cn && static_cast<fdx_cn_state*>(h->cn)->ok && static_cast<fdx_cn_state*>(h->cn)->d.cross_attention > 0; }

extern "C" int fdx_convnext_num_weights(const fdx_convnext_desc* d) {
  if (cn_validate(d)) return FDX_E_ARG;
  return 10 + d->num_layers * 13 + cn_n_cross(*d) * kCrossTensors + 4;
}

extern "C" int fdx_convnext_packed_bytes(const fdx_convnext_desc* d, size_t* bytes) {
  if (cn_validate(d) || !bytes) return FDX_E_ARG;
  CnLayout l;
  cn_layout(*d, l);
  *bytes = l.total_floats * sizeof(float);
  return FDX_OK;
}

// Canonical order = the module's state_dict order (convnext.py:170-205): input_projection.{weight,bias},
// diffusion_embedding.1.*, diffusion_embedding.3.*, conditioner_projection.0.*, conditioner_projection.2.*, then per layer:
// gamma, dwconv.{weight,bias}, norm.{weight,bias}, pwconv1.*, pwconv2.*, diffusion_step_projection.*, condition_projection.*;
// then output_projection.0.*, output_projection.2.*.
// cross_attention = n > 0: `residual_layers` of the reference is [Cross_0, Block_0 .. Block_{n-1}, Cross_1, Block_n, ...]; a cross block
// contributes, in its own state_dict order, position_scale_query, position_scale_key, positional_embedding, the 18 tensors of
// nn.TransformerDecoderLayer (declayer.hip.h), diffusion_step_projection.{weight,bias}.
extern "C" int fdx_convnext_pack(const fdx_convnext_desc* d, const float* const* w, int n, void* out, size_t bytes) {
  if (cn_validate(d)) return FDX_E_ARG;
  if (!w || !out) return fail(nullptr, FDX_E_ARG, "null pointer");
  if (n != fdx_convnext_num_weights(d)) return fail(nullptr, FDX_E_ARG, "expected %d weight tensors, got %d", fdx_convnext_num_weights(d), n);
  CnLayout l;
  cn_layout(*d, l);
  if (bytes != l.total_floats * sizeof(float)) return fail(nullptr, FDX_E_ARG, "packed size mismatch");
  float* A = static_cast<float*>(out);
  memset(A, 0, bytes);
  const int D = d->dim, H = D * d->mlp_factor, L = d->num_layers;
  int k = 0;
  pack_lin(A, l.in_proj, w[k], D, d->mel_channels, w[k + 1]); k += 2;
  pack_lin(A, l.emb1, w[k], H, D, w[k + 1]); k += 2;
  pack_lin(A, l.emb3, w[k], D, H, w[k + 1]); k += 2;
  pack_lin(A, l.cond0, w[k], H, d->condition_dim, w[k + 1]); k += 2;
  pack_lin(A, l.cond2, w[k], D, H, w[k + 1]); k += 2;
  const int NC = cn_n_cross(*d);
  for (int i = 0; i < L; ++i) {
    if (NC && i % d->cross_attention == 0) {
      const int c = i / d->cross_attention;
      const CnLayout::Cross& x = l.cross[c];
      A[x.scale_q] = w[k][0]; A[x.scale_k] = w[k + 1][0];
      if (c == 0) memcpy(A + l.pos, w[k + 2], (size_t)kCnPositions * D * sizeof(float));
      else if (memcmp(A + l.pos, w[k + 2], (size_t)kCnPositions * D * sizeof(float)) != 0)
        return fail(nullptr, FDX_E_NOIMPL, "convnext: the cross blocks' positional_embedding buffers differ (one shared table is packed)");
      k += 3;
      k += pack_declayer(A, x.dec, w + k, D, H);
      pack_lin(A, l.dsp, w[k], D, D, w[k + 1], (L + c) * D); k += 2;
    }
    memcpy(A + l.gamma[i], w[k], D * sizeof(float)); k += 1;
    memcpy(A + l.dw_w[i], w[k], (size_t)D * 7 * sizeof(float)); memcpy(A + l.dw_b[i], w[k + 1], D * sizeof(float)); k += 2;
    k += 2;   // norm.weight, norm.bias: folded into pwconv1 below
    {  // pwconv1 with the LayerNorm affine folded in: W1' = W1 diag(norm.weight), b1' = W1 norm.bias + b1 (fp64 sums, rounded once)
      const float* lw = w[k - 2]; const float* lb = w[k - 1]; const float* W1 = w[k]; const float* b1 = w[k + 1];
      std::vector<float> Wf((size_t)H * D), bf(H);
      for (int r = 0; r < H; ++r) {
        double acc = b1[r];
        for (int c = 0; c < D; ++c) { Wf[(size_t)r * D + c] = W1[(size_t)r * D + c] * lw[c]; acc += (double)W1[(size_t)r * D + c] * (double)lb[c]; }
        bf[r] = (float)acc;
      }
      pack_lin(A, l.pw1[i], Wf.data(), H, D, bf.data()); k += 2;
      for (int r = 0; r < H; ++r)            // R[r][g] = sum over group g's 32 channels of the folded weights
        for (int g = 0; g < D / kCnCh; ++g) {
          double acc = 0;
          for (int c = g * kCnCh; c < (g + 1) * kCnCh; ++c) acc += (double)Wf[(size_t)r * D + c];
          A[l.lnR[i] + (size_t)r * 16 + g] = (float)acc;
        }
      for (int r = 0; r < H; ++r) {
        double acc = 0;
        for (int c = 0; c < D; ++c) acc += (double)Wf[(size_t)r * D + c];
        A[l.lnRs[i] + r] = (float)acc;
      }
    }
    pack_lin(A, l.pw2[i], w[k], D, H, w[k + 1]);
    pack_lin(A, l.pw2w[i], w[k], D, H, w[k + 1]); k += 2;
    pack_lin(A, l.dsp, w[k], D, D, w[k + 1], i * D); k += 2;
    pack_lin(A, l.cproj, w[k], D, D, w[k + 1], i * D); k += 2;
  }
  pack_lin(A, l.out0, w[k], D, D, w[k + 1]); k += 2;
  pack_lin(A, l.out2, w[k], d->mel_channels, D, w[k + 1]); k += 2;
  return FDX_OK;
}

extern "C" int fdx_convnext_attach(fdx_handle h, const fdx_convnext_desc* d, const void* dev, size_t bytes) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (cn_validate(d)) { h->err = g_last_error; return FDX_E_ARG; }
  fdx_cn_state* S = cn(h);
  cn_layout(*d, S->l);
  if (!dev || bytes != S->l.total_floats * sizeof(float)) return fail(h, FDX_E_ARG, "packed arena size mismatch");
  S->d = *d; S->arena = static_cast<const float*>(dev); S->ok = true;
  ++h->alloc_gen;   // cached sampler graphs bake the arena address in
  h->prepared = false;
  S->pw1_16_ok = false;
  if (cn_pw1_16s() && cn_lnp() && !S->l.pw1.empty() && S->l.pw1[0].RB == 1 && (d->dim * d->mlp_factor) % 64 == 0) {
    // one-off at model load: default stream, synchronous (like the WaveNet's derived orders)
    FDX_HIP(h, hipSetDevice(h->device));
    size_t tot = 0;
    for (const auto& p : S->l.pw1) tot += 2 * packed_floats(p.n_mtiles, 1, p.cin8, 1);
    FDX_HIP(h, S->pw1_16.ensure(tot * sizeof(float), false, nullptr));
    S->pw1_off4.clear(); S->pw1_off2.clear();
    size_t c = 0;
    for (const auto& p : S->l.pw1) {
      const size_t nf = packed_floats(p.n_mtiles, 1, p.cin8, 1);
      S->pw1_off4.push_back(c); S->pw1_off2.push_back(c + nf);
      const size_t n4 = nf / 4, n2 = nf / 2;
      hipLaunchKernelGGL(k_repack16_from32rb1<4>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, nullptr, S->pw1_16.f() + c, S->arena + p.w_off, p.n_mtiles, p.cin8);
      hipLaunchKernelGGL(k_repack16_from32rb1<2>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, nullptr, S->pw1_16.f() + c + nf, S->arena + p.w_off, p.n_mtiles, p.cin8);
      c += 2 * nf;
    }
    FDX_HIP(h, hipDeviceSynchronize());
    S->pw1_16_ok = true;
  }
  return FDX_OK;
}

// ================================================================================================ prepare (hoisted conditioner path)
extern "C" int fdx_convnext_prepare(fdx_handle h, const float* cond, int B, int T, const uint8_t* cond_mask, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  fdx_cn_state* S = cn(h);
  if (!S->ok) return fail(h, FDX_E_STATE, "fdx_convnext_prepare: no weights attached");
  if (!cond || B <= 0 || T <= 0) return fail(h, FDX_E_ARG, "fdx_convnext_prepare: bad cond/B/T");
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  const auto& d = S->d;
  const auto& l = S->l;
  const float* A = S->arena;
  const int D = d.dim, H = D * d.mlp_factor, L = d.num_layers, E = d.condition_dim, M = d.mel_channels;
  const int ld = padded_ld(T, 64);
  const bool geom = B != h->B || T != h->T || h->den_kind != 1;
  h->B = B; h->T = T; h->ld = ld; h->den_kind = 1; h->den_M = M;
  auto sz = [&](int ch) { return (size_t)B * ch * ld * sizeof(float); };
  CnBufs& b = S->b;
  FDX_HIP(h, h->xin.ensure(sz(M), geom, s));
  FDX_HIP(h, h->EPS.ensure(sz(M), geom, s));
  // (N is pwconv1's B operand: the 16x16x4 family's 16 NM-column tiles may read up to 127 columns past T in the last row -- values unused, memory owned)
  FDX_HIP(h, b.X.ensure(sz(D), geom, s)); FDX_HIP(h, b.N.ensure(sz(D) + kTailPad * sizeof(float), geom, s)); FDX_HIP(h, b.H2.ensure(sz(D), geom, s));
  if (S->pw1_16_ok) {
    const int rows16 = H / 16;
    Shape16 sh{4, 4};
    const long wg44 = (long)(rows16 / 4) * B * ((T + 63) / 64);
    if (wg44 < 2 * 256) sh = pick_shape16(rows16, B, T, 12000.0 / (32.0 * ((D / 8 + 3) / 4)));
    static const int forced = [] { const char* e = getenv("FDX_CN_PW1_SHAPE"); return e ? atoi(e) : -1; }();
    if (forced > 0) sh = Shape16{forced / 10, forced % 10};
    else if (sh.NR == 4 && (long)(rows16 / 2) * B * ((T + 16 * sh.NM - 1) / (16 * sh.NM)) <= 512)
      sh.NR = 2;   // two co-resident 32-row workgroups per CU cover each other's GELU epilogue (batch 1 x 861: 32 x 112 105.2x, 64 x 112 102.9x; `r06_convnext_pw1_16s_shapes.txt`)
    if ((sh.NR != 2 && sh.NR != 4) || sh.NM < 4 || sh.NM > 8) sh = Shape16{4, 4};
    S->pw1_nr = sh.NR; S->pw1_nm = sh.NM;
  }
  FDX_HIP(h, b.G.ensure(sz(H), geom, s)); FDX_HIP(h, b.c1.ensure(sz(H), geom, s)); FDX_HIP(h, b.c2.ensure(sz(D), geom, s));
  FDX_HIP(h, b.condp.ensure(sz(E), geom, s)); FDX_HIP(h, b.CP.ensure(sz(L * D), geom, s));
  FDX_HIP(h, b.ST.ensure((size_t)B * T * 32 * sizeof(float), false, s));
  // condition = conditioner_projection(conditioner).masked_fill(cond_masks)  (convnext.py:242,247-248)
  hipLaunchKernelGGL(k_copy_rows, ew_grid(T, B * E), dim3(kEwBlock), 0, s, b.condp.f() + kHalo, (long)E * ld, ld, cond, (long)E * T, T, E, T,
                     1.f, (const uint8_t*)nullptr);
  FDX_HIP(h, gemm(A, l.cond0, B, T, b.condp.f() + kHalo, (long)E * ld, ld,
                  bias_epi(b.c1.f() + kHalo, (long)H * ld, ld, A + l.cond0.b_off, H, ACT_GELU), s));
  {
    EpiBias e = bias_epi(b.c2.f() + kHalo, (long)D * ld, ld, A + l.cond2.b_off, D, ACT_NONE);
    e.mask = cond_mask; e.mask_ld = T;
    FDX_HIP(h, gemm(A, l.cond2, B, T, b.c1.f() + kHalo, (long)H * ld, ld, e, s));
  }
  if (cond_mask) {   // PLMS evaluates the denoiser once WITHOUT masks (diffusion.py:285): keep the unmasked condition for that call
    FDX_HIP(h, b.c2raw.ensure(sz(D), geom || b.c2raw.cap < sz(D), s));
    FDX_HIP(h, gemm(A, l.cond2, B, T, b.c1.f() + kHalo, (long)H * ld, ld,
                    bias_epi(b.c2raw.f() + kHalo, (long)D * ld, ld, A + l.cond2.b_off, D, ACT_NONE), s));
  }
  const int NC = cn_n_cross(d);
  if (!NC) {
    // per-layer condition_projection(condition) for all layers in one GEMM (convnext.py:72)
    FDX_HIP(h, gemm(A, l.cproj, B, T, b.c2.f() + kHalo, (long)D * ld, ld,
                    bias_epi(b.CP.f() + kHalo, (long)L * D * ld, ld, A + l.cproj.b_off, L * D, ACT_NONE), s));
  } else {
    // cross-attention variant: the ConvNeXt blocks get no condition (convnext.py:246-250); every cross block's memory
    // condition + pos * scale_key (:137-141) is step-invariant -> its key / value projection is hoisted here
    const bool ragged = h->n_items() > 0;
    if (ragged && (B != 1 || h->items_T != T))
      return fail(h, FDX_E_ARG, "fdx_convnext_prepare: the item layout describes one row of %d frames, got a batch of %d x %d", h->items_T, B, T);
    const int Tpos = ragged ? h->items_max_len : T;
    if (Tpos > kCnPositions) return fail(h, FDX_E_ARG, "fdx_convnext_prepare: %d frames exceed the positional table (%d)", Tpos, kCnPositions);
    const int* pidx = ragged ? static_cast<const int*>(h->pidx_dev.p) : nullptr;
    FDX_HIP(h, b.QKV.ensure(sz(3 * D), geom, s)); FDX_HIP(h, b.O.ensure(sz(D), geom, s)); FDX_HIP(h, b.MEM.ensure(sz(D), geom, s));
    FDX_HIP(h, b.AP.ensure(attn_part_floats(B, T, D, ld, h->n_items()) * sizeof(float), false, s));
  FDX_HIP(h, b.AML.ensure(attn_ml_floats(B, T, h->n_items(), h->items_max_len) * sizeof(float), false, s));
    FDX_HIP(h, b.KVc.ensure(sz(NC * 2 * D), geom, s));
    for (int c = 0; c < NC; ++c) {
      const auto& x = l.cross[c];
      hipLaunchKernelGGL(k_cn_addpos, ew_grid(T, B * D), dim3(kEwBlock), 0, s, b.MEM.f() + kHalo, b.c2.f() + kHalo, (long)D * ld, ld,
                         (const float*)nullptr, 0, 0, A + l.pos, A + x.scale_k, D, T, pidx);
      FDX_HIP(h, gemm(A, x.dec.ca_kv, B, T, b.MEM.f() + kHalo, (long)D * ld, ld,
                      bias_epi(b.KVc.f() + kHalo + (size_t)c * 2 * D * ld, (long)NC * 2 * D * ld, ld, A + x.dec.ca_kv.b_off, 2 * D, ACT_NONE), s));
    }
    if (cond_mask) {   // key-padding mask of the memory: private copy (stable address for recorded graphs)
      FDX_HIP(h, b.cmask.ensure((size_t)B * T, false, s));
      FDX_HIP(h, hipMemcpyAsync(b.cmask.p, cond_mask, (size_t)B * T, hipMemcpyDeviceToDevice, s));
    }
  }
  h->cond_masked = cond_mask != nullptr;
  h->prepared = true;
  return FDX_OK;
}

// ================================================================================================ step embeddings
// SB[(l*D + c)][j] = diffusion_step_projection_l(MLP(embedding(t_j)))[c]   (convnext.py:241,64)
int fdx_cn_embed(fdx_ctx* h, const float* t_dev, int n, hipStream_t s) {
  fdx_cn_state* S = cn(h);
  const auto& d = S->d;
  const auto& l = S->l;
  const float* A = S->arena;
  const int D = d.dim, H = D * d.mlp_factor, L = d.num_layers;
  const int ldn = padded_ld(n, 64);
  CnBufs& b = S->b;
  const bool geom = ldn != b.ldn;
  b.ldn = ldn; b.n_emb = n;
  FDX_HIP(h, b.E.ensure((size_t)D * ldn * 4, geom, s));
  FDX_HIP(h, b.Hm.ensure((size_t)H * ldn * 4, geom, s));
  FDX_HIP(h, b.S0.ensure((size_t)D * ldn * 4, geom, s));
  const int NC = cn_n_cross(d);
  FDX_HIP(h, b.SB.ensure((size_t)(L + NC) * D * ldn * 4, geom, s));
  hipLaunchKernelGGL(k_step_embed, ew_grid(n, D), dim3(kEwBlock), 0, s, b.E.f() + kHalo, ldn, t_dev, n, D);
  FDX_HIP(h, gemm(A, l.emb1, 1, n, b.E.f() + kHalo, 0, ldn, bias_epi(b.Hm.f() + kHalo, 0, ldn, A + l.emb1.b_off, H, ACT_GELU), s));
  FDX_HIP(h, gemm(A, l.emb3, 1, n, b.Hm.f() + kHalo, 0, ldn, bias_epi(b.S0.f() + kHalo, 0, ldn, A + l.emb3.b_off, D, ACT_NONE), s));
  FDX_HIP(h, gemm(A, l.dsp, 1, n, b.S0.f() + kHalo, 0, ldn, bias_epi(b.SB.f() + kHalo, 0, ldn, A + l.dsp.b_off, (L + NC) * D, ACT_NONE), s));
  return FDX_OK;
}

// PLMS set-up when the batch was prepared with cond_masks: per-layer projections of the unmasked condition
int fdx_cn_plms_setup(fdx_ctx* h, hipStream_t s) {
  fdx_cn_state* S = cn(h);
  const auto& l = S->l;
  const int D = S->d.dim, L = S->d.num_layers, ld = h->ld;
  const int NC = cn_n_cross(S->d);
  if (NC) {   // cross-attention variant: keys / values of the UNMASKED condition for the one unmasked call
    const float* A = S->arena;
    FDX_HIP(h, S->b.KVc2.ensure((size_t)h->B * NC * 2 * D * ld * sizeof(float), false, s));
    for (int c = 0; c < NC; ++c) {
      const auto& x = l.cross[c];
      hipLaunchKernelGGL(k_cn_addpos, ew_grid(h->T, h->B * D), dim3(kEwBlock), 0, s, S->b.MEM.f() + kHalo, S->b.c2raw.f() + kHalo, (long)D * ld, ld,
                         (const float*)nullptr, 0, 0, A + l.pos, A + x.scale_k, D, h->T, (const int*)nullptr);
      FDX_HIP(h, gemm(A, x.dec.ca_kv, h->B, h->T, S->b.MEM.f() + kHalo, (long)D * ld, ld,
                      bias_epi(S->b.KVc2.f() + kHalo + (size_t)c * 2 * D * ld, (long)NC * 2 * D * ld, ld, A + x.dec.ca_kv.b_off, 2 * D, ACT_NONE), s));
    }
    return FDX_OK;
  }
  FDX_HIP(h, S->b.CP2.ensure((size_t)h->B * L * D * ld * sizeof(float), false, s));
  FDX_HIP(h, gemm(S->arena, l.cproj, h->B, h->T, S->b.c2raw.f() + kHalo, (long)D * ld, ld,
                  bias_epi(S->b.CP2.f() + kHalo, (long)L * D * ld, ld, S->arena + l.cproj.b_off, L * D, ACT_NONE), s));
  return FDX_OK;
}

// ================================================================================================ forward
int fdx_cn_forward_core(fdx_ctx* h, const float* xin, int col0, int sb_bs, const uint8_t* mask, float* eps_out, long o_bs, int ldo,
                        hipStream_t s, bool unmasked_cond, const EpiUniPC* fuse) {
  fdx_cn_state* S = cn(h);
  const auto& d = S->d;
  const auto& l = S->l;
  const float* A = S->arena;
  const int D = d.dim, H = D * d.mlp_factor, L = d.num_layers, M = d.mel_channels;
  const int B = h->B, T = h->T, ld = h->ld;
  CnBufs& b = S->b;
  const long bsD = (long)D * ld, bsH = (long)H * ld;
  float* X = b.X.f() + kHalo; float* N = b.N.f() + kHalo; float* G = b.G.f() + kHalo; float* H2 = b.H2.f() + kHalo;
  const float* SB = b.SB.f() + kHalo + col0;
  const int NC = cn_n_cross(d);
  const float* CP = NC ? nullptr : (unmasked_cond ? b.CP2.f() : b.CP.f()) + kHalo;
  const float* KVc = NC ? (unmasked_cond ? b.KVc2.f() : b.KVc.f()) + kHalo : nullptr;
  const uint8_t* cmask = (NC && h->cond_masked && !unmasked_cond) ? static_cast<const uint8_t*>(b.cmask.p) : nullptr;
  // pwconv2 has only D rows: 32-row tiles double the workgroup count when 64-row tiles would not fill the 256 CUs (batch 1-2 at
  // 10 s); above that the 64-row tiles' higher operand reuse wins
  const bool wide_pw2 = (long)B * ((T + 63) / 64) * ((D + 63) / 64) >= 256;
  {  // x = gelu(input_projection(x)).masked_fill(x_masks)
    EpiBias e = bias_epi(X, bsD, ld, A + l.in_proj.b_off, D, ACT_GELU);
    e.mask = mask; e.mask_ld = T;
    FDX_HIP(h, gemm(A, l.in_proj, B, T, xin, (long)M * ld, ld, e, s));
  }
  for (int i = 0; i < L; ++i) {
    if (NC && i % d.cross_attention == 0) {   // CrossAttentionBlock (convnext.py:127-152)
      const int c = i / d.cross_attention;
      const auto& x = l.cross[c];
      const bool ragged = h->n_items() > 0 && B == 1 && h->items_T == T;
      AttnItems items;
      if (ragged) { items.dev = static_cast<const int4*>(h->items_dev.p); items.host = &h->items; items.max_len = h->items_max_len; }
      hipLaunchKernelGGL(k_cn_addpos, ew_grid(T, B * D), dim3(kEwBlock), 0, s, X, X, bsD, ld, SB + (size_t)(L + c) * D * b.ldn, b.ldn, sb_bs,
                         A + l.pos, A + x.scale_q, D, T, ragged ? static_cast<const int*>(h->pidx_dev.p) : (const int*)nullptr);
      const DecScratch sc{b.QKV.f() + kHalo, b.O.f() + kHalo, G, b.AP.f() + kHalo, b.AML.f()};
      FDX_HIP(h, run_declayer(A, x.dec, B, T, D, H, ld, X, KVc + (size_t)c * 2 * D * ld, (long)NC * 2 * D * ld, sc, mask, cmask, s, nullptr,
                              nullptr, 1, 0, items));
    }
    const dim3 grid((T + 63) / 64, D / kCnCh, B);
    hipLaunchKernelGGL(k_dwconv_stats, grid, dim3(256), 0, s, X, bsD, ld, T, l.dil[i], D, CP ? CP + (size_t)i * D * ld : nullptr, (long)L * D * ld,
                       SB + (size_t)i * D * b.ldn, b.ldn, sb_bs, A + l.dw_w[i], A + l.dw_b[i], mask, N, b.ST.f(), cn_lnp() ? 0 : 1);
    {  // pwconv1 over LayerNorm(u): centring + rstd inside the GEMM, affine folded into the packed weights
      const PackedW& p = l.pw1[i];
      ConvGeom gg{B, T, p.cin8, 1, 0, 0, p.n_mtiles};
      hipEvent_t ev0 = nullptr, ev1 = nullptr;   // fdx_prof_*: the denoiser's dominant kernel
      if (S->pw1_16_ok)
        h->prof.note(PROF_CN_PWCONV1, "convgemm16s_kernel<EpiBiasAct16S<%d>, %d, %d, PRE_LNP> (v_mfma_f32_16x16x4_f32; %d x %d split-K workgroup tile, LayerNorm folded in, GELU epilogue; %ld workgroups)",
                     S->pw1_nm, S->pw1_nr, S->pw1_nm, 16 * S->pw1_nr, 16 * S->pw1_nm, (long)B * ((T + 16 * S->pw1_nm - 1) / (16 * S->pw1_nm)) * (H / (16 * S->pw1_nr)));
      else
        h->prof.note(PROF_CN_PWCONV1, "convgemm_kernel<%d, true, %s, EpiBias> (v_mfma_f32_32x32x2_f32; %d x 64 split-K workgroup tile, LayerNorm folded in, GELU epilogue; %ld workgroups)",
                     p.RB, cn_lnp() ? "PRE_LNP" : "PRE_LN", 32 * p.RB, (long)B * ((T + 63) / 64) * p.n_mtiles);
      h->prof.take(PROF_CN_PWCONV1, 2.0 * (double)H * D * (double)B * T, ev0, ev1);
      const float4* Wp = reinterpret_cast<const float4*>(A + p.w_off);
      const EpiBias eb = bias_epi(G, bsH, ld, A + p.b_off, H, ACT_GELU);
      if (S->pw1_16_ok) {
        const int NRs = S->pw1_nr, NMs = S->pw1_nm;
        const ConvGeom g4{B, T, p.cin8, 1, 0, 0, H / 64}, g2{B, T, p.cin8, 1, 0, 0, H / 32};
        const void* W4 = S->pw1_16.f() + S->pw1_off4[i];
        const void* W2 = S->pw1_16.f() + S->pw1_off2[i];
        hipError_t e = hipErrorInvalidValue;
#define FDX_PW1_SHAPE(NR_, NM_)                                                                                                        \
  if (NRs == NR_ && NMs == NM_) {                                                                                                      \
    const EpiBiasAct16S<NM_> ea{G, bsH, ld, A + p.b_off, ACT_GELU};                                                                    \
    e = launch_convgemm16s<EpiBiasAct16S<NM_>, NR_, NM_, PRE_LNP>(NR_ == 4 ? g4 : g2, NR_ == 4 ? W4 : W2, N, bsD, ld, ea, s, ev0, ev1, \
                                                                 b.ST.f(), A + l.lnRs[i], D / kCnCh, 1e-6f);                           \
  }
        FDX_PW1_SHAPE(4, 4) FDX_PW1_SHAPE(4, 5) FDX_PW1_SHAPE(4, 6) FDX_PW1_SHAPE(4, 7) FDX_PW1_SHAPE(4, 8)
        FDX_PW1_SHAPE(2, 4) FDX_PW1_SHAPE(2, 5) FDX_PW1_SHAPE(2, 6) FDX_PW1_SHAPE(2, 7) FDX_PW1_SHAPE(2, 8)
#undef FDX_PW1_SHAPE
        FDX_HIP(h, e);
      } else if (cn_lnp()) {
        if (p.RB == 1) FDX_HIP(h, (launch_convgemm<1, true, PRE_LNP, EpiBias>(gg, Wp, N, bsD, ld, 1.f, eb, s, ev0, ev1, b.ST.f(), A + l.lnRs[i], D / kCnCh, 1e-6f)));
        else FDX_HIP(h, (launch_convgemm<2, true, PRE_LNP, EpiBias>(gg, Wp, N, bsD, ld, 1.f, eb, s, ev0, ev1, b.ST.f(), A + l.lnRs[i], D / kCnCh, 1e-6f)));
      } else {
        if (p.RB == 1) FDX_HIP(h, (launch_convgemm<1, true, PRE_LN, EpiBias>(gg, Wp, N, bsD, ld, 1.f, eb, s, ev0, ev1, b.ST.f(), A + l.lnR[i], D / kCnCh, 1e-6f)));
        else FDX_HIP(h, (launch_convgemm<2, true, PRE_LN, EpiBias>(gg, Wp, N, bsD, ld, 1.f, eb, s, ev0, ev1, b.ST.f(), A + l.lnR[i], D / kCnCh, 1e-6f)));
      }
    }
    EpiScaleRes e{};
    e.X = X; e.bs = bsD; e.ld = ld; e.bias = A + l.pw2[i].b_off; e.gamma = A + l.gamma[i]; e.M = D; e.mask = mask; e.mask_ld = T;
    FDX_HIP(h, gemm(A, wide_pw2 ? l.pw2w[i] : l.pw2[i], B, T, G, bsH, ld, e, s));
  }
  FDX_HIP(h, gemm(A, l.out0, B, T, X, bsD, ld, bias_epi(H2, bsD, ld, A + l.out0.b_off, D, ACT_GELU), s));
  if (fuse) {   // UniPC: eps is consumed in the epilogue (corrector + the next step's predictor), bit-identical to the separate launch
    EpiUniPC e = *fuse;
    e.bias = A + l.out2.b_off; e.M = M; e.mask = mask; e.mask_ld = T;
    FDX_HIP(h, gemm(A, l.out2, B, T, H2, bsD, ld, e, s));
  } else {
    EpiBias e = bias_epi(eps_out, o_bs, ldo, A + l.out2.b_off, M, ACT_NONE);
    e.mask = mask; e.mask_ld = T;
    e.tight = ldo != ld;
    FDX_HIP(h, gemm(A, l.out2, B, T, H2, bsD, ld, e, s));
  }
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

extern "C" int fdx_convnext_forward(fdx_handle h, const float* x, const float* t, int n_t, const uint8_t* x_mask, float* eps,
                                    fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  fdx_cn_state* S = cn(h);
  if (!S->ok || !h->prepared || h->den_kind != 1) return fail(h, FDX_E_STATE, "fdx_convnext_forward: call attach + prepare first");
  if (!x || !t || !eps) return fail(h, FDX_E_ARG, "fdx_convnext_forward: null pointer");
  if (n_t != 1 && n_t != h->B) return fail(h, FDX_E_ARG, "diffusion_step must have 1 or B=%d entries, got %d", h->B, n_t);
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  const int M = S->d.mel_channels, B = h->B, T = h->T, ld = h->ld;
  if (int rc = fdx_cn_embed(h, t, n_t, s)) return rc;
  hipLaunchKernelGGL(k_copy_rows, ew_grid(T, B * M), dim3(kEwBlock), 0, s, h->xin.f() + kHalo, (long)M * ld, ld, x, (long)M * T, T, M, T, 1.f,
                     (const uint8_t*)nullptr);
  return fdx_cn_forward_core(h, h->xin.f() + kHalo, 0, n_t == 1 ? 0 : 1, x_mask, eps, (long)M * T, T, s, false);
}
