// tfdec.hip -- the TransformerDecoderDenoiser: fish_diffusion/modules/convnext.py:263-379, registered as
// DENOISERS "TransformerDecoderDenoiser" (archs/diffsinger/diffusions/builder.py:13).  12 x torch.nn.TransformerDecoderLayer
// (post-norm, 8 heads, GELU feed-forward, batch_first) between 1x1-conv projections; same call contract as the WaveNet, so
// the sampler loop of wavenet.hip drives it through fdx_td_embed / fdx_td_forward_core.
//
// Activations stay channel-major [B][D][ld] (frames contiguous), so every Linear / 1x1 conv is a convgemm launch on the packed
// weights, and attention reads Q / K / V as rows of those matrices:
//   S^T[key][query] = K^T Q    (MFMA 32x32x2: A = K rows read along frames, B = Q rows kept in registers for the whole launch)
//   online softmax over the key axis = over accumulator registers of one lane (+ one cross-half exchange): no row shuffles
//   O^T[d][query]  += V P^T    (A = the V tile staged through LDS, B = the S^T accumulators used in place: the MFMA k-pair
//                               of step r is exactly the key pair (acc_row(r, 0), acc_row(r, 1)) the two lane halves hold)
// One workgroup = 64 queries x 1 head; its 4 waves split the key tiles and merge their (max, sum, O) triples through LDS.
// Hoisted per utterance batch: condition_projection(conditioner) + positional embedding (fdx_tfdec_prepare); per sampler run:
// the step-embedding MLP for all timesteps.  Per call the memory = mask(C0 + step) is rebuilt (one elementwise launch) and each
// layer projects its own K / V from it -- exactly the reference's order of operations (no algebraic shortcuts through softmax).
#include "common.hip.h"
#include "elementwise.hip.h"
#include "gemmplan.hip.h"

#include <cmath>

using namespace fdx;

namespace {

constexpr int kHeads = 8;   // nn.TransformerDecoderLayer(nhead=8), convnext.py:300

struct TdLayer {
  PackedW sa_in, sa_out, ca_q, ca_kv, ca_out, lin1, lin2;
  size_t n1w, n1b, n2w, n2b, n3w, n3b;
};
struct TdLayout {
  PackedW in0, in2, emb1, emb3, cond0, cond2, out0, out2;
  size_t pos = 0;            // positional_embedding [n_positions][D], raw
  size_t scale_q = 0, scale_k = 0;
  std::vector<TdLayer> layers;
  size_t total_floats = 0;
};

int td_validate(const fdx_tfdec_desc* d) {
  if (!d) return fail(nullptr, FDX_E_ARG, "null tfdec desc");
  if (d->dim != 128 && d->dim != 256 && d->dim != 512)
    return fail(nullptr, FDX_E_ARG, "tfdec: dim must be 128, 256 or 512 (8 heads of 16 / 32 / 64), got %d", d->dim);
  if (d->mlp_factor < 1 || d->mlp_factor > 8) return fail(nullptr, FDX_E_ARG, "tfdec: mlp_factor out of range");
  if (d->mel_channels <= 0 || d->mel_channels % 8 || d->condition_dim <= 0 || d->condition_dim % 8)
    return fail(nullptr, FDX_E_ARG, "tfdec: mel_channels and condition_dim must be multiples of 8");
  if (d->num_layers <= 0 || d->n_positions <= 0) return fail(nullptr, FDX_E_ARG, "tfdec: num_layers / n_positions must be positive");
  return FDX_OK;
}

void td_layout(const fdx_tfdec_desc& d, TdLayout& l) {
  const int D = d.dim, H = D * d.mlp_factor;
  size_t cur = 0;
  l.scale_q = cur; cur += 64;
  l.scale_k = cur; cur += 64;
  l.pos = cur; cur += (size_t)round_up(d.n_positions * D, 64);
  l.in0 = plan64(cur, H, d.mel_channels);
  l.in2 = plan32(cur, D, H);
  l.emb1 = plan64(cur, H, D);
  l.emb3 = plan64(cur, D, H);
  l.cond0 = plan64(cur, H, d.condition_dim);
  l.cond2 = plan32(cur, D, H);
  l.layers.assign(d.num_layers, TdLayer{});
  for (auto& y : l.layers) {
    y.sa_in = plan64(cur, 3 * D, D);
    y.sa_out = plan32(cur, D, D);
    y.ca_q = plan32(cur, D, D);
    y.ca_kv = plan64(cur, 2 * D, D);
    y.ca_out = plan32(cur, D, D);
    y.lin1 = plan64(cur, H, D);
    y.lin2 = plan32(cur, D, H);
    for (size_t* p : {&y.n1w, &y.n1b, &y.n2w, &y.n2b, &y.n3w, &y.n3b}) { *p = cur; cur += round_up(D, 64); }
  }
  l.out0 = plan32(cur, D, D);
  l.out2 = plan64(cur, d.mel_channels, D);
  l.total_floats = cur;
}

// ------------------------------------------------------------------------------------------------ elementwise
// X[b][c][t] = masked ? 0 : X[b][c][t] + pos[t][c] * scale        (convnext.py:344-346,356-357 / :348,353)
__global__ void k_td_addpos(float* __restrict__ X, long bs, int ld, const float* __restrict__ pos, const float* __restrict__ scale,
                            const uint8_t* __restrict__ mask, int D, int T) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const int b = blockIdx.y / D, c = blockIdx.y - b * D;
  const long o = b * bs + (long)c * ld + t;
  const float v = X[o] + pos[(long)t * D + c] * scale[0];
  X[o] = (mask && mask[(long)b * T + t]) ? 0.f : v;
}

// mem[b][c][t] = masked ? 0 : C0[b][c][t] + step[c]                (convnext.py:353,359-360)
__global__ void k_td_mem(float* __restrict__ mem, const float* __restrict__ C0, long bs, int ld, const float* __restrict__ S0,
                         int s_ld, int s_bs, const uint8_t* __restrict__ mask, int D, int T) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const int b = blockIdx.y / D, c = blockIdx.y - b * D;
  const long o = b * bs + (long)c * ld + t;
  const float v = C0[o] + S0[(long)c * s_ld + b * s_bs];
  mem[o] = (mask && mask[(long)b * T + t]) ? 0.f : v;
}

// LayerNorm over channels, in place: 8 frames x 32 channel groups per workgroup (T/8 workgroups: latency-bound, wants many),
// values in registers, two-pass statistics
constexpr int kLnFr = 8, kLnCg = 32, kLnCpt = 16;
__global__ __launch_bounds__(256) void k_td_layernorm(float* __restrict__ X, long bs, int ld, const float* __restrict__ w,
                                                      const float* __restrict__ bia, int D, int T, float eps) {
  __shared__ float red[kLnCg][kLnFr];
  const int tc = threadIdx.x & (kLnFr - 1), cg = threadIdx.x / kLnFr;
  const int b = blockIdx.y, t = blockIdx.x * kLnFr + tc;
  const int cpt = D / kLnCg;
  const bool live = t < T;
  float* xb = X + b * bs + (live ? t : T - 1);
  float u[kLnCpt];
  float s1 = 0.f;
#pragma unroll
  for (int ci = 0; ci < kLnCpt; ++ci) {
    u[ci] = 0.f;
    if (ci < cpt) { u[ci] = xb[(long)(cg * cpt + ci) * ld]; s1 += u[ci]; }
  }
  auto total = [&](float v) {
    __syncthreads();
    red[cg][tc] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kLnCg; ++g) s += red[g][tc];
    return s;
  };
  const float mean = total(s1) / (float)D;
  float s2 = 0.f;
#pragma unroll
  for (int ci = 0; ci < kLnCpt; ++ci)
    if (ci < cpt) { const float dl = u[ci] - mean; s2 += dl * dl; }
  const float rstd = 1.f / sqrtf(total(s2) / (float)D + eps);
  if (!live) return;
#pragma unroll
  for (int ci = 0; ci < kLnCpt; ++ci)
    if (ci < cpt) { const int c = cg * cpt + ci; xb[(long)c * ld] = (u[ci] - mean) * rstd * w[c] + bia[c]; }
}

// ------------------------------------------------------------------------------------------------ attention
struct AttnArgs {
  const float* Q; long q_bs; int ldq;     // head h, channel d, frame t at Q[b*q_bs + (h*DH + d)*ldq + t]
  const float* K; long k_bs; int ldk;
  const float* V; long v_bs; int ldv;
  float* O; long o_bs; int ldo;
  const uint8_t* kmask;                   // [B][Tk] bytes, 1 = key ignored (key_padding_mask), or null
  int Tq, Tk;
  float scale;                            // 1 / sqrt(DH)
};

// NQ = 32-query blocks per workgroup: 2 (64 queries) when that already fills the chip, 1 to double the workgroup count
template <int DH, int NQ>
__global__ __launch_bounds__(256) void k_attn(AttnArgs a) {
  constexpr int KS = DH / 2;                 // MFMA k-steps of the score product
  constexpr int RBD = (DH + 31) / 32;        // 32-row blocks of O^T
  constexpr int VLD = 65;                    // padded row of the staged V tile
  constexpr int NBLK = RBD * NQ;
  constexpr int LDS_F = (4 * DH * VLD > 4 * NBLK * 16 * 64 + 4 * 2 * NQ * 64) ? 4 * DH * VLD : 4 * NBLK * 16 * 64 + 4 * 2 * NQ * 64;
  __shared__ float lds[LDS_F];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, n = lane & 31;
  const int q0 = blockIdx.x * (32 * NQ), h = blockIdx.y, b = blockIdx.z;
  const float* Qh = a.Q + b * a.q_bs + (long)h * DH * a.ldq;
  const float* Kh = a.K + b * a.k_bs + (long)h * DH * a.ldk;
  const float* Vh = a.V + b * a.v_bs + (long)h * DH * a.ldv;
  const float NEG = -__builtin_inff();

  // B operand of the score product: this lane's slice of Q, for the whole launch
  float qreg[KS][NQ];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int nb = 0; nb < NQ; ++nb) qreg[ks][nb] = Qh[(long)(2 * ks + half) * a.ldq + min(q0 + nb * 32 + n, a.Tq - 1)];

  f32x16 o[RBD][NQ];
#pragma unroll
  for (int x = 0; x < RBD; ++x)
#pragma unroll
    for (int nb = 0; nb < NQ; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[x][nb][r] = 0.f;
  float m[NQ], l[NQ];
#pragma unroll
  for (int nb = 0; nb < NQ; ++nb) { m[nb] = NEG; l[nb] = 0.f; }
  float* vt = lds + wave * DH * VLD;

  // The K and V operands of a tile are fetched into registers one tile ahead: right after the score MFMAs have consumed the
  // current ones, so that their fabric latency runs behind the softmax and the second product instead of in front of the first.
  const int n_kt = (a.Tk + 63) / 64;
  float kreg[KS][2], vreg[DH];
  auto fetch = [&](int kt) {
    const int k0 = kt * 64;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) kreg[ks][rb] = Kh[(long)(2 * ks + half) * a.ldk + min(k0 + rb * 32 + n, a.Tk - 1)];
#pragma unroll
    for (int d = 0; d < DH; ++d) vreg[d] = Vh[(long)d * a.ldv + min(k0 + lane, a.Tk - 1)];
  };
  // (not for 64-query x 64-channel workgroups: their accumulators leave no room for a second operand set in 512 VGPRs)
  constexpr bool PF = !(DH == 64 && NQ == 2);
  if (PF && wave < n_kt) fetch(wave);
  for (int kt = wave; kt < n_kt; kt += 4) {
    const int k0 = kt * 64;
    // ---- stage this tile of V (coalesced rows) for the second product
    if constexpr (PF) {
#pragma unroll
      for (int d = 0; d < DH; ++d) vt[d * VLD + lane] = vreg[d];
    } else {
#pragma unroll 8
      for (int d = 0; d < DH; ++d) vt[d * VLD + lane] = Vh[(long)d * a.ldv + min(k0 + lane, a.Tk - 1)];
    }
    // ---- S^T = K^T Q
    f32x16 s[2][NQ];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < NQ; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[rb][nb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float ak[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        ak[rb] = PF ? kreg[ks][rb] : Kh[(long)(2 * ks + half) * a.ldk + min(k0 + rb * 32 + n, a.Tk - 1)];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < NQ; ++nb) s[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[rb], qreg[ks][nb], s[rb][nb], 0, 0, 0);
    }
    if (PF && kt + 4 < n_kt) fetch(kt + 4);
    // ---- scale, key mask (padding keys and the tile overhang), online softmax over the key axis
    float mx[NQ];
#pragma unroll
    for (int nb = 0; nb < NQ; ++nb) mx[nb] = NEG;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + rb * 32 + acc_row(r, half);
        const bool ok = key < a.Tk && !(a.kmask && a.kmask[(long)b * a.Tk + min(key, a.Tk - 1)]);
#pragma unroll
        for (int nb = 0; nb < NQ; ++nb) {
          const float v = ok ? s[rb][nb][r] * a.scale : NEG;
          s[rb][nb][r] = v;
          mx[nb] = fmaxf(mx[nb], v);
        }
      }
#pragma unroll
    for (int nb = 0; nb < NQ; ++nb) {
      mx[nb] = fmaxf(mx[nb], __shfl_xor(mx[nb], 32));
      const float m_new = fmaxf(m[nb], mx[nb]);
      const float m_use = m_new == NEG ? 0.f : m_new;      // every key so far masked: keep exp() finite, all weights 0
      const float alpha = expf(m[nb] - m_use);
      float sum = 0.f;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __expf(s[rb][nb][r] - m_use);
          s[rb][nb][r] = p;
          sum += p;
        }
      sum += __shfl_xor(sum, 32);
      l[nb] = l[nb] * alpha + sum;
      m[nb] = m_new;
#pragma unroll
      for (int x = 0; x < RBD; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[x][nb][r] *= alpha;
    }
    // ---- O^T += V P^T : the k-pair of step (rb, r) is the key pair the two lane halves hold in s[rb][.][r]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = rb * 32 + acc_row(r, half);
#pragma unroll
        for (int x = 0; x < RBD; ++x) {
          const int d = x * 32 + n;
          const float av = d < DH ? vt[d * VLD + kl] : 0.f;
#pragma unroll
          for (int nb = 0; nb < NQ; ++nb) o[x][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s[rb][nb][r], o[x][nb], 0, 0, 0);
        }
      }
  }

  // ---- merge the 4 waves' (m, l, O) through LDS; wave w finishes accumulator rows r = 4w .. 4w+3 of every block
  __syncthreads();
  float* ob = lds;                                   // [wave][blk][r][lane]
  float* ml = lds + 4 * NBLK * 16 * 64;              // [wave][{m,l}][nb][lane]
#pragma unroll
  for (int x = 0; x < RBD; ++x)
#pragma unroll
    for (int nb = 0; nb < NQ; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[((wave * NBLK + x * NQ + nb) * 16 + r) * 64 + lane] = o[x][nb][r];
#pragma unroll
  for (int nb = 0; nb < NQ; ++nb) {
    ml[((wave * 2 + 0) * NQ + nb) * 64 + lane] = m[nb];
    ml[((wave * 2 + 1) * NQ + nb) * 64 + lane] = l[nb];
  }
  __syncthreads();
#pragma unroll
  for (int nb = 0; nb < NQ; ++nb) {
    float mw[4], M = NEG, L = 0.f, wg[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = ml[((w * 2 + 0) * NQ + nb) * 64 + lane]; M = fmaxf(M, mw[w]); }
    const float M_use = M == NEG ? 0.f : M;
#pragma unroll
    for (int w = 0; w < 4; ++w) { wg[w] = expf(mw[w] - M_use); L += ml[((w * 2 + 1) * NQ + nb) * 64 + lane] * wg[w]; }
    const int q = q0 + nb * 32 + n;
    if (q >= a.Tq) continue;
#pragma unroll
    for (int x = 0; x < RBD; ++x)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr;
        const int d = x * 32 + acc_row(r, half);
        if (d >= DH) continue;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += ob[((w * NBLK + x * NQ + nb) * 16 + r) * 64 + lane] * wg[w];
        a.O[b * a.o_bs + (long)(h * DH + d) * a.ldo + q] = acc / L;
      }
  }
}

template <int NQ>
hipError_t launch_attn_nq(int DH, const AttnArgs& a, int B, hipStream_t s) {
  const dim3 grid((a.Tq + 32 * NQ - 1) / (32 * NQ), kHeads, B), blk(256);
  if (DH == 64) hipLaunchKernelGGL((k_attn<64, NQ>), grid, blk, 0, s, a);
  else if (DH == 32) hipLaunchKernelGGL((k_attn<32, NQ>), grid, blk, 0, s, a);
  else hipLaunchKernelGGL((k_attn<16, NQ>), grid, blk, 0, s, a);
  return hipGetLastError();
}
hipError_t launch_attn(int DH, const AttnArgs& a, int B, hipStream_t s) {
  // 64-query workgroups only when they already give every CU one (B * heads * T/64 >= 256); else 32-query ones
  if ((long)B * kHeads * ((a.Tq + 63) / 64) >= 256) return launch_attn_nq<2>(DH, a, B, s);
  return launch_attn_nq<1>(DH, a, B, s);
}

struct TdBufs {
  DevBuf X, QKV, KV, O, G, Hin, H2, mem, C0, condp, c1, cmask;
  DevBuf E, Hm, S0;
  int ldn = 0;
};

}  // namespace

struct fdx_td_state {
  bool ok = false;
  fdx_tfdec_desc d{};
  TdLayout l;
  const float* arena = nullptr;
  TdBufs b;
};

static fdx_td_state* td(fdx_ctx* h) {
  if (!h->td) h->td = new fdx_td_state();
  return static_cast<fdx_td_state*>(h->td);
}
void fdx_td_free(void* p) { delete static_cast<fdx_td_state*>(p); }

extern "C" int fdx_tfdec_num_weights(const fdx_tfdec_desc* d) {
  if (td_validate(d)) return FDX_E_ARG;
  return 3 + 12 + d->num_layers * 18 + 4;
}

extern "C" int fdx_tfdec_packed_bytes(const fdx_tfdec_desc* d, size_t* bytes) {
  if (td_validate(d) || !bytes) return FDX_E_ARG;
  TdLayout l;
  td_layout(*d, l);
  *bytes = l.total_floats * sizeof(float);
  return FDX_OK;
}

// Canonical order = the module's state_dict order (convnext.py:275-313): position_scale_query, position_scale_key,
// positional_embedding, input_projection.{0,2}.{weight,bias}, diffusion_embedding.{1,3}.*, condition_projection.{0,2}.*, per layer
// self_attn.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias}, multihead_attn.(same four), linear1.*, linear2.*,
// norm1.*, norm2.*, norm3.*; then output_projection.{0,2}.*.
extern "C" int fdx_tfdec_pack(const fdx_tfdec_desc* d, const float* const* w, int n, void* out, size_t bytes) {
  if (td_validate(d)) return FDX_E_ARG;
  if (!w || !out) return fail(nullptr, FDX_E_ARG, "null pointer");
  if (n != fdx_tfdec_num_weights(d)) return fail(nullptr, FDX_E_ARG, "expected %d weight tensors, got %d", fdx_tfdec_num_weights(d), n);
  TdLayout l;
  td_layout(*d, l);
  if (bytes != l.total_floats * sizeof(float)) return fail(nullptr, FDX_E_ARG, "packed size mismatch");
  float* A = static_cast<float*>(out);
  memset(A, 0, bytes);
  const int D = d->dim, H = D * d->mlp_factor, M = d->mel_channels, E = d->condition_dim;
  int k = 0;
  A[l.scale_q] = w[k++][0];
  A[l.scale_k] = w[k++][0];
  memcpy(A + l.pos, w[k++], (size_t)d->n_positions * D * sizeof(float));
  pack_lin(A, l.in0, w[k], H, M, w[k + 1]); k += 2;
  pack_lin(A, l.in2, w[k], D, H, w[k + 1]); k += 2;
  pack_lin(A, l.emb1, w[k], H, D, w[k + 1]); k += 2;
  pack_lin(A, l.emb3, w[k], D, H, w[k + 1]); k += 2;
  pack_lin(A, l.cond0, w[k], H, E, w[k + 1]); k += 2;
  pack_lin(A, l.cond2, w[k], D, H, w[k + 1]); k += 2;
  for (auto& y : l.layers) {
    pack_lin(A, y.sa_in, w[k], 3 * D, D, w[k + 1]); k += 2;
    pack_lin(A, y.sa_out, w[k], D, D, w[k + 1]); k += 2;
    pack_lin(A, y.ca_q, w[k], D, D, w[k + 1]);                                        // in_proj rows [0, D): the query projection
    pack_lin(A, y.ca_kv, w[k] + (size_t)D * D, 2 * D, D, w[k + 1] + D); k += 2;       // rows [D, 3D): key and value
    pack_lin(A, y.ca_out, w[k], D, D, w[k + 1]); k += 2;
    pack_lin(A, y.lin1, w[k], H, D, w[k + 1]); k += 2;
    pack_lin(A, y.lin2, w[k], D, H, w[k + 1]); k += 2;
    for (size_t off : {y.n1w, y.n1b, y.n2w, y.n2b, y.n3w, y.n3b}) memcpy(A + off, w[k++], D * sizeof(float));
  }
  pack_lin(A, l.out0, w[k], D, D, w[k + 1]); k += 2;
  pack_lin(A, l.out2, w[k], M, D, w[k + 1]); k += 2;
  return FDX_OK;
}

extern "C" int fdx_tfdec_attach(fdx_handle h, const fdx_tfdec_desc* d, const void* dev, size_t bytes) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (td_validate(d)) { h->err = g_last_error; return FDX_E_ARG; }
  fdx_td_state* S = td(h);
  td_layout(*d, S->l);
  if (!dev || bytes != S->l.total_floats * sizeof(float)) return fail(h, FDX_E_ARG, "packed arena size mismatch");
  S->d = *d; S->arena = static_cast<const float*>(dev); S->ok = true;
  ++h->alloc_gen;   // cached sampler graphs bake the arena address in
  h->prepared = false;
  return FDX_OK;
}

// ================================================================================================ prepare (hoisted condition path)
extern "C" int fdx_tfdec_prepare(fdx_handle h, const float* cond, int B, int T, const uint8_t* cond_mask, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  fdx_td_state* S = td(h);
  if (!S->ok) return fail(h, FDX_E_STATE, "fdx_tfdec_prepare: no weights attached");
  if (!cond || B <= 0 || T <= 0) return fail(h, FDX_E_ARG, "fdx_tfdec_prepare: bad cond/B/T");
  const auto& d = S->d;
  if (T > d.n_positions) return fail(h, FDX_E_ARG, "fdx_tfdec_prepare: %d frames exceed the positional table (%d)", T, d.n_positions);
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  const auto& l = S->l;
  const float* A = S->arena;
  const int D = d.dim, H = D * d.mlp_factor, E = d.condition_dim, M = d.mel_channels;
  const int ld = padded_ld(T, 64);
  const bool geom = B != h->B || T != h->T || h->den_kind != 2;
  h->B = B; h->T = T; h->ld = ld; h->den_kind = 2; h->den_M = M;
  auto sz = [&](int ch) { return (size_t)B * ch * ld * sizeof(float); };
  TdBufs& b = S->b;
  FDX_HIP(h, h->xin.ensure(sz(M), geom, s));
  FDX_HIP(h, h->EPS.ensure(sz(M), geom, s));
  FDX_HIP(h, b.X.ensure(sz(D), geom, s)); FDX_HIP(h, b.QKV.ensure(sz(3 * D), geom, s)); FDX_HIP(h, b.KV.ensure(sz(2 * D), geom, s));
  FDX_HIP(h, b.O.ensure(sz(D), geom, s)); FDX_HIP(h, b.G.ensure(sz(H), geom, s)); FDX_HIP(h, b.Hin.ensure(sz(H), geom, s));
  FDX_HIP(h, b.H2.ensure(sz(D), geom, s)); FDX_HIP(h, b.mem.ensure(sz(D), geom, s)); FDX_HIP(h, b.C0.ensure(sz(D), geom, s));
  FDX_HIP(h, b.condp.ensure(sz(E), geom, s)); FDX_HIP(h, b.c1.ensure(sz(H), geom, s));
  // C0 = condition_projection(conditioner) + positional_embedding[:T] * position_scale_key   (convnext.py:348,353-357)
  hipLaunchKernelGGL(k_copy_rows, ew_grid(T, B * E), dim3(kEwBlock), 0, s, b.condp.f() + kHalo, (long)E * ld, ld, cond, (long)E * T, T, E, T,
                     1.f, (const uint8_t*)nullptr);
  FDX_HIP(h, gemm(A, l.cond0, B, T, b.condp.f() + kHalo, (long)E * ld, ld,
                  bias_epi(b.c1.f() + kHalo, (long)H * ld, ld, A + l.cond0.b_off, H, ACT_GELU), s));
  FDX_HIP(h, gemm(A, l.cond2, B, T, b.c1.f() + kHalo, (long)H * ld, ld,
                  bias_epi(b.C0.f() + kHalo, (long)D * ld, ld, A + l.cond2.b_off, D, ACT_NONE), s));
  hipLaunchKernelGGL(k_td_addpos, ew_grid(T, B * D), dim3(kEwBlock), 0, s, b.C0.f() + kHalo, (long)D * ld, ld, A + l.pos, A + l.scale_k,
                     (const uint8_t*)nullptr, D, T);
  h->cond_masked = cond_mask != nullptr;
  if (cond_mask) {   // private copy: the memory mask is applied on every call (and dropped for PLMS' one unmasked call)
    FDX_HIP(h, b.cmask.ensure((size_t)B * T, false, s));
    FDX_HIP(h, hipMemcpyAsync(b.cmask.p, cond_mask, (size_t)B * T, hipMemcpyDeviceToDevice, s));
  }
  FDX_HIP(h, hipGetLastError());
  h->prepared = true;
  return FDX_OK;
}

// ================================================================================================ step embeddings
int fdx_td_embed(fdx_ctx* h, const float* t_dev, int n, hipStream_t s) {
  fdx_td_state* S = td(h);
  const auto& l = S->l;
  const float* A = S->arena;
  const int D = S->d.dim, H = D * S->d.mlp_factor;
  const int ldn = padded_ld(n, 64);
  TdBufs& b = S->b;
  const bool geom = ldn != b.ldn;
  b.ldn = ldn;
  FDX_HIP(h, b.E.ensure((size_t)D * ldn * 4, geom, s));
  FDX_HIP(h, b.Hm.ensure((size_t)H * ldn * 4, geom, s));
  FDX_HIP(h, b.S0.ensure((size_t)D * ldn * 4, geom, s));
  hipLaunchKernelGGL(k_step_embed, ew_grid(n, D), dim3(kEwBlock), 0, s, b.E.f() + kHalo, ldn, t_dev, n, D);
  FDX_HIP(h, gemm(A, l.emb1, 1, n, b.E.f() + kHalo, 0, ldn, bias_epi(b.Hm.f() + kHalo, 0, ldn, A + l.emb1.b_off, H, ACT_GELU), s));
  FDX_HIP(h, gemm(A, l.emb3, 1, n, b.Hm.f() + kHalo, 0, ldn, bias_epi(b.S0.f() + kHalo, 0, ldn, A + l.emb3.b_off, D, ACT_NONE), s));
  return FDX_OK;
}

// ================================================================================================ forward
int fdx_td_forward_core(fdx_ctx* h, const float* xin, int col0, int sb_bs, const uint8_t* mask, float* eps_out, long o_bs, int ldo,
                        hipStream_t s, bool unmasked_cond) {
  fdx_td_state* S = td(h);
  const auto& d = S->d;
  const auto& l = S->l;
  const float* A = S->arena;
  const int D = d.dim, H = D * d.mlp_factor, M = d.mel_channels, DH = D / kHeads;
  const int B = h->B, T = h->T, ld = h->ld;
  TdBufs& b = S->b;
  const long bsD = (long)D * ld, bsH = (long)H * ld;
  float* X = b.X.f() + kHalo; float* QKV = b.QKV.f() + kHalo; float* KV = b.KV.f() + kHalo; float* O = b.O.f() + kHalo;
  float* G = b.G.f() + kHalo; float* Hin = b.Hin.f() + kHalo; float* H2 = b.H2.f() + kHalo; float* mem = b.mem.f() + kHalo;
  const uint8_t* cmask = (h->cond_masked && !unmasked_cond) ? static_cast<const uint8_t*>(b.cmask.p) : nullptr;
  const dim3 blk(kEwBlock);
  // x = input_projection(x)^T + pos * scale_q, masked                                         (convnext.py:343-346,356-357)
  FDX_HIP(h, gemm(A, l.in0, B, T, xin, (long)M * ld, ld, bias_epi(Hin, bsH, ld, A + l.in0.b_off, H, ACT_GELU), s));
  FDX_HIP(h, gemm(A, l.in2, B, T, Hin, bsH, ld, bias_epi(X, bsD, ld, A + l.in2.b_off, D, ACT_NONE), s));
  hipLaunchKernelGGL(k_td_addpos, ew_grid(T, B * D), blk, 0, s, X, bsD, ld, A + l.pos, A + l.scale_q, mask, D, T);
  // memory = mask(C0 + diffusion_step)                                                          (:353,359-360)
  hipLaunchKernelGGL(k_td_mem, ew_grid(T, B * D), blk, 0, s, mem, b.C0.f() + kHalo, bsD, ld, b.S0.f() + kHalo + col0, b.ldn, sb_bs, cmask, D, T);
  const dim3 ln_grid((T + kLnFr - 1) / kLnFr, B);
  auto residual = [&](const PackedW& p, const float* in, long in_bs) {   // X += W in + b
    EpiScaleRes e{};
    e.X = X; e.bs = bsD; e.ld = ld; e.bias = A + p.b_off; e.gamma = nullptr; e.M = D; e.mask = nullptr; e.mask_ld = T;
    return gemm(A, p, B, T, in, in_bs, ld, e, s);
  };
  AttnArgs at{};
  at.O = O; at.o_bs = bsD; at.ldo = ld; at.Tq = T; at.Tk = T; at.scale = 1.f / sqrtf((float)DH);
  for (const auto& y : l.layers) {
    // ---- self-attention block: x = norm1(x + out_proj(attn(in_proj(x))))
    FDX_HIP(h, gemm(A, y.sa_in, B, T, X, bsD, ld, bias_epi(QKV, 3 * bsD, ld, A + y.sa_in.b_off, 3 * D, ACT_NONE), s));
    at.Q = QKV; at.q_bs = 3 * bsD; at.ldq = ld;
    at.K = QKV + (size_t)D * ld; at.k_bs = 3 * bsD; at.ldk = ld;
    at.V = QKV + (size_t)2 * D * ld; at.v_bs = 3 * bsD; at.ldv = ld;
    at.kmask = mask;
    FDX_HIP(h, launch_attn(DH, at, B, s));
    FDX_HIP(h, residual(y.sa_out, O, bsD));
    hipLaunchKernelGGL(k_td_layernorm, ln_grid, dim3(256), 0, s, X, bsD, ld, A + y.n1w, A + y.n1b, D, T, 1e-5f);
    // ---- cross-attention block: x = norm2(x + out_proj(attn(q(x), k(memory), v(memory))))
    FDX_HIP(h, gemm(A, y.ca_q, B, T, X, bsD, ld, bias_epi(QKV, 3 * bsD, ld, A + y.ca_q.b_off, D, ACT_NONE), s));
    FDX_HIP(h, gemm(A, y.ca_kv, B, T, mem, bsD, ld, bias_epi(KV, 2 * bsD, ld, A + y.ca_kv.b_off, 2 * D, ACT_NONE), s));
    at.K = KV; at.k_bs = 2 * bsD; at.V = KV + (size_t)D * ld; at.v_bs = 2 * bsD;
    at.kmask = cmask;
    FDX_HIP(h, launch_attn(DH, at, B, s));
    FDX_HIP(h, residual(y.ca_out, O, bsD));
    hipLaunchKernelGGL(k_td_layernorm, ln_grid, dim3(256), 0, s, X, bsD, ld, A + y.n2w, A + y.n2b, D, T, 1e-5f);
    // ---- feed-forward block: x = norm3(x + linear2(gelu(linear1(x))))
    FDX_HIP(h, gemm(A, y.lin1, B, T, X, bsD, ld, bias_epi(G, bsH, ld, A + y.lin1.b_off, H, ACT_GELU), s));
    FDX_HIP(h, residual(y.lin2, G, bsH));
    hipLaunchKernelGGL(k_td_layernorm, ln_grid, dim3(256), 0, s, X, bsD, ld, A + y.n3w, A + y.n3b, D, T, 1e-5f);
  }
  FDX_HIP(h, gemm(A, l.out0, B, T, X, bsD, ld, bias_epi(H2, bsD, ld, A + l.out0.b_off, D, ACT_GELU), s));
  {
    EpiBias e = bias_epi(eps_out, o_bs, ldo, A + l.out2.b_off, M, ACT_NONE);
    e.mask = mask; e.mask_ld = T;
    e.tight = ldo != ld;
    FDX_HIP(h, gemm(A, l.out2, B, T, H2, bsD, ld, e, s));
  }
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

extern "C" int fdx_tfdec_forward(fdx_handle h, const float* x, const float* t, int n_t, const uint8_t* x_mask, float* eps,
                                 fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  fdx_td_state* S = td(h);
  if (!S->ok || !h->prepared || h->den_kind != 2) return fail(h, FDX_E_STATE, "fdx_tfdec_forward: call attach + prepare first");
  if (!x || !t || !eps) return fail(h, FDX_E_ARG, "fdx_tfdec_forward: null pointer");
  if (n_t != 1 && n_t != h->B) return fail(h, FDX_E_ARG, "diffusion_step must have 1 or B=%d entries, got %d", h->B, n_t);
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  const int M = S->d.mel_channels, B = h->B, T = h->T, ld = h->ld;
  if (int rc = fdx_td_embed(h, t, n_t, s)) return rc;
  hipLaunchKernelGGL(k_copy_rows, ew_grid(T, B * M), dim3(kEwBlock), 0, s, h->xin.f() + kHalo, (long)M * ld, ld, x, (long)M * T, T, M, T, 1.f,
                     (const uint8_t*)nullptr);
  return fdx_td_forward_core(h, h->xin.f() + kHalo, 0, n_t == 1 ? 0 : 1, x_mask, eps, (long)M * T, T, s, false);
}
