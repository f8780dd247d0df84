// declayer.hip.h -- one `torch.nn.TransformerDecoderLayer(d_model, nhead=8, dim_feedforward, activation="gelu", batch_first=True)`
// (post-norm; eval mode) on channel-major activations [B][D][ld], shared by the two denoisers that contain such layers:
//   tfdec.hip     TransformerDecoderDenoiser        fish_diffusion/modules/convnext.py:263-379  (12 layers)
//   convnext.hip  ConvNext(cross_attention=True)    fish_diffusion/modules/convnext.py:95-152   (a CrossAttentionBlock every 5th layer)
// Every Linear is a convgemm launch on packed weights; attention is the fp32 MFMA flash kernel below (S^T = K^T Q so that the
// softmax axis runs over the accumulator registers of one lane; P^T feeds the second product in place; see tfdec.hip's header).
#pragma once
#include "common.hip.h"
#include "convgemm16s.hip.h"
#include "elementwise.hip.h"
#include "gemmplan.hip.h"

#include <cmath>
#include <type_traits>

namespace fdx {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TdLayer {
  PackedW sa_in, sa_out, ca_q, ca_kv, ca_out, lin1, lin2;
  size_t n1w, n1b, n2w, n2b, n3w, n3b;
  // LayerNorms folded into their consumers (run_declayer_ln): row sums [round_up(rows, 64)] of the folded weights (PRE_LNP)
  bool folded = false;
  size_t sa_in_R = 0, ca_q_R = 0, lin1_R = 0;
};

// Tile height of the self-attention in-projection and of linear1: 32-row tiles (RB = 1).  Round 6, tools/tdbench.py at T = 861: with 64-row
// tiles the two GEMMs are 336 / 448 workgroups -- 1.3 / 1.75 per CU, so the CUs that hold two set the time; 32-row tiles (672 / 896 workgroups,
// four co-resident per CU) balance: 1903 -> 1836 -> 1791 us per denoiser call at batch 1, 9540 -> 9130 at batch 8.
// FDX_TD_SAIN_RB / FDX_TD_LIN1_RB = 2: the 64-row tiles of rounds 4-5 (A/B; decides the arena layout, read once per process).
inline int declayer_sa_in_rb() {
  static const int v = [] { const char* e = getenv("FDX_TD_SAIN_RB"); const int k = e ? atoi(e) : 0; return k == 2 ? 2 : 1; }();
  return v;
}
inline int declayer_lin1_rb() {
  static const int v = [] { const char* e = getenv("FDX_TD_LIN1_RB"); const int k = e ? atoi(e) : 0; return k == 2 ? 2 : 1; }();
  return v;
}
inline void plan_declayer(size_t& cur, TdLayer& y, int D, int H, bool folded = false) {
  y.sa_in = declayer_sa_in_rb() == 1 ? plan32(cur, 3 * D, D) : plan64(cur, 3 * D, D);
  y.sa_out = plan32(cur, D, D);
  y.ca_q = plan32(cur, D, D);
  y.ca_kv = plan64(cur, 2 * D, D);
  y.ca_out = plan32(cur, D, D);
  y.lin1 = declayer_lin1_rb() == 1 ? plan32(cur, H, D) : plan64(cur, H, D);
  y.lin2 = plan32(cur, D, H);
  for (size_t* p : {&y.n1w, &y.n1b, &y.n2w, &y.n2b, &y.n3w, &y.n3b}) { *p = cur; cur += round_up(D, 64); }
  y.folded = folded;
  if (folded) {
    y.sa_in_R = cur; cur += (size_t)round_up(3 * D, 64);
    y.ca_q_R = cur; cur += (size_t)round_up(D, 64);
    y.lin1_R = cur; cur += (size_t)round_up(H, 64);
  }
}

// A Linear behind a LayerNorm, with the norm's affine part folded in (as convnext.hip does for pwconv1):
//   W (((u - mean) rstd) w + b) + b1  =  rstd ((W diag(w)) u - mean R) + (W b + b1);   R[r] = row sum of W diag(w)
// (fp64 sums, rounded once).  lnw == null: packed as it is, R untouched.
inline void pack_lin_ln(float* A, const PackedW& p, size_t R_off, const float* W, int rows, int cin, const float* bias, const float* lnw,
                        const float* lnb) {
  if (!lnw) { pack_lin(A, p, W, rows, cin, bias); return; }
  std::vector<float> Wf((size_t)rows * cin), bf(rows);
  for (int r = 0; r < rows; ++r) {
    double acc = bias ? bias[r] : 0.0;
    for (int c = 0; c < cin; ++c) { Wf[(size_t)r * cin + c] = W[(size_t)r * cin + c] * lnw[c]; acc += (double)W[(size_t)r * cin + c] * (double)lnb[c]; }
    bf[r] = (float)acc;
  }
  pack_lin(A, p, Wf.data(), rows, cin, bf.data());
  for (int r = 0; r < rows; ++r) {
    double acc = 0;
    for (int c = 0; c < cin; ++c) acc += (double)Wf[(size_t)r * cin + c];
    A[R_off + r] = (float)acc;
  }
}

// 18 tensors in nn.TransformerDecoderLayer's state_dict order: self_attn.{in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias},
// multihead_attn.(same four), linear1.{weight, bias}, linear2.{weight, bias}, norm1.{weight, bias}, norm2.*, norm3.*.  Returns 18.
// Folded layers: ca_q carries norm1, linear1 norm2, and the self-attention in-projection the PREVIOUS layer's norm3 (`prev_n3w/b`; null for the
// first layer, whose input is a plain tensor) -- this layer's norm3 goes into whatever consumes the layer's output.
inline int pack_declayer(float* A, const TdLayer& y, const float* const* w, int D, int H, const float* prev_n3w = nullptr,
                         const float* prev_n3b = nullptr) {
  const bool f = y.folded;
  const float* n1w = f ? w[12] : nullptr; const float* n1b = f ? w[13] : nullptr;
  const float* n2w = f ? w[14] : nullptr; const float* n2b = f ? w[15] : nullptr;
  int k = 0;
  pack_lin_ln(A, y.sa_in, y.sa_in_R, w[k], 3 * D, D, w[k + 1], f ? prev_n3w : nullptr, prev_n3b); k += 2;
  pack_lin(A, y.sa_out, w[k], D, D, w[k + 1]); k += 2;
  pack_lin_ln(A, y.ca_q, y.ca_q_R, w[k], D, D, w[k + 1], n1w, n1b);                  // in_proj rows [0, D): the query projection
  pack_lin(A, y.ca_kv, w[k] + (size_t)D * D, 2 * D, D, w[k + 1] + D); k += 2;       // rows [D, 3D): key and value
  pack_lin(A, y.ca_out, w[k], D, D, w[k + 1]); k += 2;
  pack_lin_ln(A, y.lin1, y.lin1_R, w[k], H, D, w[k + 1], n2w, n2b); k += 2;
  pack_lin(A, y.lin2, w[k], D, H, w[k + 1]); k += 2;
  for (size_t off : {y.n1w, y.n1b, y.n2w, y.n2b, y.n3w, y.n3b}) memcpy(A + off, w[k++], D * sizeof(float));
  return k;
}

// X[b][c][t] = masked ? 0 : X[b][c][t] + pos[t][c] * scale        (convnext.py:344-346,356-357 / :348,353)
// pidx (exact-ragged rows, fdx_sampler_set_items): the frame's position inside its own item; null = its column
__global__ void k_td_addpos(float* __restrict__ X, long bs, int ld, const float* __restrict__ pos, const float* __restrict__ scale,
                            const uint8_t* __restrict__ mask, int D, int T, const int* __restrict__ pidx) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const int b = blockIdx.y / D, c = blockIdx.y - b * D;
  const long o = b * bs + (long)c * ld + t;
  const float v = X[o] + pos[(long)(pidx ? pidx[t] : t) * D + c] * scale[0];
  X[o] = (mask && mask[(long)b * T + t]) ? 0.f : v;
}


// LayerNorm over channels, in place: FR frames x (256 / FR) channel groups per workgroup, values in registers, two-pass statistics.
// The kernel is latency-bound (1.8 MB in, 1.8 MB out at batch 1): what matters is how wide the row segments are that a wave touches
// per load (FR * 4 bytes) against how many workgroups there are (T / FR); 11.3 us per launch at T = 861, D = 512 before the loads
// became unconditional, 5.5 us after (FR = 16).
template <int FR>
__global__ __launch_bounds__(256) void k_td_layernorm(float* __restrict__ X, long bs, int ld, const float* __restrict__ w,
                                                      const float* __restrict__ bia, int D, int T, float eps) {
  constexpr int CG = 256 / FR, CPT = 512 / CG;       // channel groups; channels per thread at the largest D (512)
  __shared__ float red[CG][2][FR];                   // per channel group and frame: the group's mean and centred sum of squares
  const int tc = threadIdx.x & (FR - 1), cg = threadIdx.x / FR;
  const int b = blockIdx.y, t = blockIdx.x * FR + tc;
  const int cpt = D / CG;
  const bool live = t < T;
  float* xb = X + b * bs + (live ? t : T - 1);
  // (unconditional loads from a clamped row: behind `if (ci < cpt)` hipcc waits for each load before it issues the next --
  // the kernel's time was proportional to the channels per thread: 11.3 us at 16)
  float u[CPT];
#pragma unroll
  for (int ci = 0; ci < CPT; ++ci) u[ci] = xb[(long)(cg * cpt + min(ci, cpt - 1)) * ld];
  // (the affine parameters are fetched HERE, with the frame: behind the reduction's barrier -- where they are used, and where hipcc leaves
  // a load it may not move across a barrier -- they cost the kernel one more memory round trip at its very end)
  float wv[CPT], bv[CPT];
#pragma unroll
  for (int ci = 0; ci < CPT; ++ci) { const int c = cg * cpt + min(ci, cpt - 1); wv[ci] = w[c]; bv[ci] = bia[c]; }
  // Two-pass statistics of the thread's own channels (exact), then ONE exchange: every thread publishes (mean_g, M2_g) of its group and
  // combines the CG groups of its frame in a fixed order (Chan et al.: M2 = sum M2_g + n_g sum (mean_g - mean)^2 -- products of centred
  // quantities only).  Round 4 ran two shuffle + LDS reductions back to back (mean, then the centred squares): two more barriers.
  float s1 = 0.f;
#pragma unroll
  for (int ci = 0; ci < CPT; ++ci) s1 += ci < cpt ? u[ci] : 0.f;
  const float mean_g = s1 / (float)cpt;
  float m2_g = 0.f;
#pragma unroll
  for (int ci = 0; ci < CPT; ++ci) { const float dl = u[ci] - mean_g; m2_g += ci < cpt ? dl * dl : 0.f; }
  red[cg][0][tc] = mean_g;
  red[cg][1][tc] = m2_g;
  __syncthreads();
  float msum = 0.f, m2 = 0.f;
#pragma unroll
  for (int g = 0; g < CG; ++g) { msum += red[g][0][tc]; m2 += red[g][1][tc]; }
  const float mean = msum / (float)CG;
  float dev2 = 0.f;
#pragma unroll
  for (int g = 0; g < CG; ++g) { const float dl = red[g][0][tc] - mean; dev2 += dl * dl; }
  const float rstd = 1.f / sqrtf((m2 + (float)cpt * dev2) / (float)D + eps);
  if (!live) return;
#pragma unroll
  for (int ci = 0; ci < CPT; ++ci)
    if (ci < cpt) xb[(long)(cg * cpt + ci) * ld] = (u[ci] - mean) * rstd * wv[ci] + bv[ci];
}
inline int ln_frames() {   // FDX_LN_FR = 4 | 8 | 16 | 32 (A/B)
  static const int v = [] { const char* e = getenv("FDX_LN_FR"); const int k = e ? atoi(e) : 0; return (k == 4 || k == 8 || k == 16 || k == 32) ? k : 16; }();   // 16: 2272 us per call; 8: 2294; 4: 2332; 32: 2379
  return v;
}
inline void launch_layernorm(float* X, long bs, int ld, const float* w, const float* bia, int B, int D, int T, hipStream_t s) {
  const int fr = (D % 64) ? 8 : ln_frames();          // FR = 4 -> 64 channel groups: D must divide by them
  const dim3 grid((T + fr - 1) / fr, B);
  if (fr == 4) hipLaunchKernelGGL(k_td_layernorm<4>, grid, dim3(256), 0, s, X, bs, ld, w, bia, D, T, 1e-5f);
  else if (fr == 32) hipLaunchKernelGGL(k_td_layernorm<32>, grid, dim3(256), 0, s, X, bs, ld, w, bia, D, T, 1e-5f);
  else if (fr == 16) hipLaunchKernelGGL(k_td_layernorm<16>, grid, dim3(256), 0, s, X, bs, ld, w, bia, D, T, 1e-5f);
  else hipLaunchKernelGGL(k_td_layernorm<8>, grid, dim3(256), 0, s, X, bs, ld, w, bia, D, T, 1e-5f);
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
  // query-split kernel only (k_attn_qs): key range split over `ksplit` workgroups; split s leaves its un-normalised O^T at P + s * p_split
  // (addressed like O) and its (max, sum) per query at ML[((s * B + b) * kHeads + h) * 2 + {0, 1}][TqR]
  float* P; long p_split;
  float* ML; int TqR;
  int ksplit, B;
  // exact-ragged rows: grid.z walks the items of ONE row; item i = columns [items[i].x, + items[i].y) is its own attention problem (Tq = Tk =
  // its length) and splits its keys as a batch-1 run of it alone would (attn_ksplit_of(1, len)): `ksplit` is then the largest over the items
  const int4* items; int ks_force;
#ifdef FDX_ATTN_TRACE
  unsigned long long* trace;              // [workgroup][wave][8] shader-clock stamps (tools/ubench/attnqs.hip only)
#endif
};
#ifdef FDX_ATTN_TRACE
#define FDX_ATTN_STAMP(k) do { if (a.trace && lane == 0) a.trace[(((long)blockIdx.z * gridDim.x + blockIdx.x) * 4 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FDX_ATTN_STAMP(k) do { } while (0)
#endif
// FDX_ATTN_TRACE == 2: the phases INSIDE the second tile instead of whole tiles (slots 3..7: tile start, scores issued, softmax done, second
// product issued, next tile staged + barrier); the order of the phases is pinned, so this build is a little slower than the product's
#if defined(FDX_ATTN_TRACE) && FDX_ATTN_TRACE == 2
#define FDX_ATTN_FINE(k) do { __builtin_amdgcn_sched_barrier(0); if (kt == 1) FDX_ATTN_STAMP(k); __builtin_amdgcn_sched_barrier(0); } while (0)
#define FDX_ATTN_COARSE(k) do { } while (0)
#else
#define FDX_ATTN_FINE(k) do { } while (0)
#define FDX_ATTN_COARSE(k) FDX_ATTN_STAMP(k)
#endif

#ifndef FDX_ATTN_PF
#define FDX_ATTN_PF 1     // fragment prefetch inside the two products (0 = round 5's loops, kept for the A/B in tools/ubench/attnqs.hip)
#endif
// ------------------------------------------------------------------------------------------------ attention, query-split (round 5)
// Round 4's kernel (k_attn, removed; git history and profiles/r05_attention_ubench_*.txt hold its numbers) gave a workgroup 32 queries and let
// its four waves split the KEY tiles: every wave streamed its own K / V tiles from global memory into registers, a workgroup read the head's whole
// K and V (441 KB at T = 861) for 7 MFLOP, and the launch sat on the global_load -> VGPR path (16 of the 18 B/clk/CU it delivers): 29 us, 33 % of
// the fp32 MFMA roof (187 us at batch 8).  Measured here: 22.4 us + 5.4 us of combine at batch 1, 125 us at batch 8.  The four waves of a workgroup own
// 32 QUERIES each and share every K / V tile through LDS (one cooperative 16-byte-per-lane fetch per tile, double-buffered, one barrier per
// tile), so a tile is fetched once per 128 queries instead of once per 32; the chip is filled by splitting the KEY range over `ksplit`
// workgroups (flash-decoding): each leaves (max, sum, un-normalised O^T) and k_attn_combine folds them in a fixed order.  Operand bytes per
// workgroup: 110 KB of K / V + 32 KB of Q at T = 861 (was 441 + 8).  The softmax runs in base 2 (log2 e folded into the query scale).
//   grid.x = kHeads * n_qblocks * ksplit with the HEAD in the low bits of the linear id: workgroup i lands on XCD i % 8, so head h's Q / K / V
//   live in ONE XCD's L2 (8 heads, 8 XCDs) instead of in all eight.
//   LDS per buffer: K tile as [ks][rb][half][n] (exactly the lane order of the score product's A operand: conflict-free ds_read_b32 pairs),
//   V tile as [d][65] (the second product reads it transposed: stride 65 over the lanes of a half = 32 distinct banks).
template <int DH>
__global__ __launch_bounds__(256) void k_attn_qs(AttnArgs a) {
  constexpr int KS = DH / 2;                 // MFMA k-steps of the score product
  constexpr int RBD = (DH + 31) / 32;        // 32-row blocks of O^T
  constexpr int VLD = 65;
  constexpr int KT = KS * 128, VT = DH * VLD;          // floats per staged K / V tile (64 keys)
  constexpr int UPT = DH / 16;               // 16-byte fetch units per thread per operand tile (DH rows x 16 units / 256 threads)
  constexpr int PB = KS >= 8 ? 4 : KS / 2;   // k-steps per prefetched fragment batch of the score product
  __shared__ float lds[2 * (KT + VT)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, n = lane & 31;
  const int lin = blockIdx.x;
  const int h = lin % kHeads, j = lin / kHeads;
  const int split = j % a.ksplit, qb = j / a.ksplit;
  int b = blockIdx.z, slot = blockIdx.z, col0 = 0, ksplit = a.ksplit;
  if (a.items) {                             // exact-ragged row: this workgroup's item (everything below sees it as a batch-1 problem at col0)
    const int4 it = a.items[blockIdx.z];
    b = 0; col0 = it.x; a.Tq = a.Tk = it.y;
    ksplit = attn_ksplit_of(1, it.y, it.y, a.ks_force);
    if (split >= ksplit || qb * 128 >= it.y) return;
  }
  const int q0 = qb * 128 + wave * 32;
  const int units = (a.Tk + 31) >> 5;        // 32-key units, dealt to the splits in balanced runs
  const int u0 = (int)((long)units * split / ksplit), u1 = (int)((long)units * (split + 1) / ksplit);
  const int kbeg = u0 * 32, kend = min(u1 * 32, a.Tk);
  const int n_kt = (u1 - u0 + 1) >> 1;       // 64-key tiles; the last one may hold 32 keys
  const float* Qh = a.Q + b * a.q_bs + (long)h * DH * a.ldq + col0;
  const float* Kh = a.K + b * a.k_bs + (long)h * DH * a.ldk + col0;
  const float* Vh = a.V + b * a.v_bs + (long)h * DH * a.ldv + col0;
  const float NEG = -__builtin_inff();
  const bool has_mask = a.kmask != nullptr && !a.items;     // (an item's keys are its own frames: no padding inside)
  const uint8_t* mrow = has_mask ? a.kmask + (long)b * a.Tk : reinterpret_cast<const uint8_t*>(Kh);

  // B operand of the score product: this lane's slice of Q for the whole launch, pre-multiplied by log2(e) / sqrt(DH).  The loads go out
  // FIRST and the first K / V tile right behind them (pinned: hipcc sank the Q loads below the first barrier, two fabric round trips in a row)
  FDX_ATTN_STAMP(0);
  float qreg[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qreg[ks] = Qh[(long)(2 * ks + half) * a.ldq + min(q0 + n, a.Tq - 1)];
  __builtin_amdgcn_sched_barrier(0);

  f32x16 o[RBD];
#pragma unroll
  for (int x = 0; x < RBD; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[x][r] = 0.f;
  float m = NEG, l = 0.f;

  // cooperative fetch of one 64-key tile: unit u = i * 256 + tid -> row d = u / 16, keys 4 (u % 16) .. + 3 (16 lanes = one 256-byte row segment)
  f32x4 kld[UPT], vld[UPT];
  unsigned mnext = 0;
  auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = i * 256 + tid, d = u >> 4, c = (u & 15) * 4;
      kld[i] = *reinterpret_cast<const f32x4*>(Kh + (long)d * a.ldk + k0 + c);
      vld[i] = *reinterpret_cast<const f32x4*>(Vh + (long)d * a.ldv + k0 + c);
    }
    mnext = mrow[(unsigned)min(k0 + lane, a.Tk - 1)];
  };
  auto stage = [&](float* buf, int k0) __attribute__((always_inline)) {
    float* vb = buf;           // V tile first: its fragment reads then reach every key from two lane bases with 8-bit dword offsets, and the K
    float* kb = buf + VT;      // tile (VT = 65 x 64 dwords behind) every fragment from ONE base with ds_read2st64's 64-dword offset units
    const bool tail = k0 + 64 > a.Tk;          // columns past Tk are row padding: whatever they hold must not reach 0 * V
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = i * 256 + tid, d = u >> 4, c = (u & 15) * 4;
      *reinterpret_cast<f32x4*>(kb + (d >> 1) * 128 + (c >> 5) * 64 + (d & 1) * 32 + (c & 31)) = kld[i];
      f32x4 v = vld[i];
      if (tail) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k0 + c + e < a.Tk) ? v[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) vb[d * VLD + c + e] = v[e];
    }
  };

  fetch(kbeg);
  __builtin_amdgcn_sched_barrier(0);
  const float qs = a.scale * 1.44269504088896340736f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qreg[ks] *= qs;
  FDX_ATTN_STAMP(1);
  stage(lds, kbeg);
  unsigned mreg = mnext;
  __syncthreads();
  FDX_ATTN_STAMP(2);
  for (int kt = 0; kt < n_kt; ++kt) {
    const int k0 = kbeg + kt * 64;
    const bool more = kt + 1 < n_kt;
    float* vb = lds + (kt & 1) * (KT + VT);
    float* kb = vb + VT;
    FDX_ATTN_FINE(3);
    if (more) fetch(k0 + 64);                      // in flight behind this tile's two products
    // keys this tile must ignore (past this split's range / past Tk / key padding), as a wave-uniform bit set
    const unsigned long long badm = __ballot((k0 + lane >= kend) || (has_mask && mreg != 0));
    const bool both = k0 + 32 < kend;             // the second 32 keys of the tile exist (workgroup-uniform)
    // ---- S^T = K^T Q
    f32x16 s[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[rb][r] = 0.f;
#if FDX_ATTN_PF
    // fragment reads run ONE BATCH (PB k-steps) ahead of the MFMAs that consume them, and the order is pinned: left to itself hipcc issues a
    // batch's ds_reads right in front of its first MFMA and the matrix pipe idles for an LDS round trip every 8 MFMAs (4.92 k cycles per
    // product against 4096 of MFMA issue)
    if (both) {
      float fa[2][2 * PB];
#pragma unroll
      for (int i = 0; i < PB; ++i) { fa[0][2 * i] = kb[i * 128 + lane]; fa[0][2 * i + 1] = kb[i * 128 + 64 + lane]; }
#pragma unroll
      for (int bt = 0; bt < KS / PB; ++bt) {
        if (bt + 1 < KS / PB) {
#pragma unroll
          for (int i = 0; i < PB; ++i) {
            fa[(bt + 1) & 1][2 * i] = kb[((bt + 1) * PB + i) * 128 + lane];
            fa[(bt + 1) & 1][2 * i + 1] = kb[((bt + 1) * PB + i) * 128 + 64 + lane];
          }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[bt & 1][2 * i], qreg[bt * PB + i], s[0], 0, 0, 0);
          s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[bt & 1][2 * i + 1], qreg[bt * PB + i], s[1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, PB, 0);          // the first batch's reads
#pragma unroll
      for (int bt = 0; bt < KS / PB; ++bt)
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          if (bt + 1 < KS / PB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 ds_read2st64 of the next batch
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                         // 2 MFMAs of this one
        }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[ks * 128 + lane], qreg[ks], s[0], 0, 0, 0);
    }
#else
    if (both) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float a0 = kb[ks * 128 + lane], a1 = kb[ks * 128 + 64 + lane];
        s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qreg[ks], s[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, qreg[ks], s[1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[ks * 128 + lane], qreg[ks], s[0], 0, 0, 0);
    }
#endif
    FDX_ATTN_FINE(4);
    // ---- key mask + online softmax over the key axis (accumulator registers of one lane + one cross-half exchange)
    if (badm != 0ull) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = rb * 32 + acc_row(r, 0);          // this element's key bit in the low lane half; + 4 in the high half
          const unsigned long long lanes = (((badm >> c) & 1) ? 0x00000000FFFFFFFFull : 0ull) | (((badm >> (c + 4)) & 1) ? 0xFFFFFFFF00000000ull : 0ull);
          const bool bad = __builtin_amdgcn_inverse_ballot_w64(lanes);
          s[rb][r] = bad ? NEG : s[rb][r];
        }
    }
    // The reference maximum m moves only when some query of the wave sees a score more than 2^8 above it (always on the first tile): between
    // moves the weights are 2^(s - m) <= 256 and O^T / l need no rescaling -- same sum, same normalisation, 16 fewer VALU passes over the
    // accumulators per tile (fp32 VALU work is additive to the fp32 MFMAs).
    float mx = NEG;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[rb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__ballot(mx > m + 8.f) != 0ull) {
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - (m_new == NEG ? 0.f : m_new));   // m = -inf (nothing seen yet): 0
      l *= alpha;
      m = m_new;
      if (kt > 0) {
#pragma unroll
        for (int x = 0; x < RBD; ++x)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[x][r] *= alpha;
      }
    }
    const float m_use = m == NEG ? 0.f : m;      // every key so far masked: keep exp2() finite, all weights 0
    float sum = 0.f;
#if FDX_ATTN_PF
    // ---- weights + O^T += V P^T, interleaved.  The k-pair of step (rb, r) is the key pair the two lane halves hold in s[rb][r].  V fragments
    // are read ONE BATCH (4 key pairs) ahead of their MFMAs; only the first 32 keys' exponentials stand in front of the product, the second
    // 32 keys' ride in the VALU slots between its first 16 RBD MFMAs (same operations on the same values in the same order as the plain loops
    // below: identical bits).  The order is pinned -- left alone hipcc reads each batch right in front of its first MFMA
    {
      float fv[2][4 * RBD];
      auto ldv2 = [&](float* f, int bt, int i0) __attribute__((always_inline)) {     // key pairs i0, i0 + 1 of batch bt (adjacent keys: one ds_read2 per x)
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
          const int kl = (bt >> 2) * 32 + acc_row((bt & 3) * 4 + i, half);
#pragma unroll
          for (int x = 0; x < RBD; ++x) {
            const int d = x * 32 + n;
            f[i * RBD + x] = d < DH ? vb[d * VLD + kl] : 0.f;
          }
        }
      };
      auto ldv = [&](float* f, int bt) __attribute__((always_inline)) { ldv2(f, bt, 0); ldv2(f, bt, 2); };
      ldv(fv[0], 0);                              // in flight behind the first exponentials
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[0][r] - m_use);
        s[0][r] = p;
        sum += p;
      }
      FDX_ATTN_FINE(5);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int bt = 0; bt < 4; ++bt)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {           // half batches, fenced: 2 RBD MFMAs behind RBD reads and two keys' exponentials
          ldv2(fv[(bt + 1) & 1], bt + 1, 2 * hb);
#pragma unroll
          for (int i = 2 * hb; i < 2 * hb + 2; ++i) {   // (a half tile's second 32 keys are all masked: -inf -> weight 0, as in the plain loop)
            const float p = __builtin_amdgcn_exp2f(s[1][bt * 4 + i] - m_use);
            s[1][bt * 4 + i] = p;
            sum += p;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 2 * hb; i < 2 * hb + 2; ++i)
#pragma unroll
            for (int x = 0; x < RBD; ++x) o[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[bt & 1][i * RBD + x], s[0][bt * 4 + i], o[x], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (both) {
#pragma unroll
        for (int bt = 4; bt < 8; ++bt) {
          if (bt + 1 < 8) ldv(fv[(bt + 1) & 1], bt + 1);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int x = 0; x < RBD; ++x) o[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[bt & 1][i * RBD + x], s[1][(bt & 3) * 4 + i], o[x], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2 * RBD; ++i) {
            if (bt + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
        }
      }
      sum += __shfl_xor(sum, 32);
      l += sum;
    }
#else
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[rb][r] - m_use);
        s[rb][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    l += sum;
    FDX_ATTN_FINE(5);
    // ---- O^T += V P^T : the k-pair of step (rb, r) is the key pair the two lane halves hold in s[rb][r]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      if (rb == 1 && !both) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = rb * 32 + acc_row(r, half);
#pragma unroll
        for (int x = 0; x < RBD; ++x) {
          const int d = x * 32 + n;
          const float av = d < DH ? vb[d * VLD + kl] : 0.f;
          o[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s[rb][r], o[x], 0, 0, 0);
        }
      }
    }
#endif
    FDX_ATTN_FINE(6);
    if (more) {
      stage(lds + ((kt + 1) & 1) * (KT + VT), k0 + 64);
      mreg = mnext;
    }
    __syncthreads();
    FDX_ATTN_FINE(7);
    if (kt < 3) FDX_ATTN_COARSE(3 + kt);
  }
  FDX_ATTN_COARSE(6);

  const int q = q0 + n;
  if (q >= a.Tq) return;
  if (ksplit == 1) {
    const float rl = 1.f / l;
#pragma unroll
    for (int x = 0; x < RBD; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = x * 32 + acc_row(r, half);
        if (d < DH) a.O[b * a.o_bs + (long)(h * DH + d) * a.ldo + col0 + q] = o[x][r] * rl;
      }
    return;
  }
  float* P = a.P + split * a.p_split + b * a.o_bs + col0;
#pragma unroll
  for (int x = 0; x < RBD; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = x * 32 + acc_row(r, half);
      if (d < DH) P[(long)(h * DH + d) * a.ldo + q] = o[x][r];
    }
  if (half == 0) {
    float* ml = a.ML + (((long)split * a.B + slot) * kHeads + h) * 2 * a.TqR;
    ml[q] = m;
    ml[a.TqR + q] = l;
  }
}

// O[c][q] = sum_s 2^(m_s - M) P_s[c][q] / sum_s 2^(m_s - M) l_s,  M = max_s m_s  -- splits folded in index order (deterministic).
// One thread = one query x 4 channels; KSP is a template parameter so that all 4 KSP operand loads are unconditional and in flight
// together (with `if (s < ksplit)` around each load hipcc waited for every load in turn: 64 dependent round trips, 21 us per launch).
// (scalar arguments, the 14 dwords the dispatcher preloads first: a by-value AttnArgs is fetched from the kernarg segment, one fabric round trip in
// front of a kernel that is three round trips long)
template <int DH, int KSP>
__global__ __launch_bounds__(256) void k_attn_combine(const float* __restrict__ P, const float* __restrict__ ML, float* __restrict__ O, long o_bs,
                                                      long p_split, int ldo, int TqR, int Tq, int B, const int4* __restrict__ items, int ks_force) {
  constexpr int CH = DH / 16;                 // 16-channel chunks per head
  const int q = blockIdx.x * 64 + (threadIdx.x & 63), cg = threadIdx.x >> 6;
  const int h = blockIdx.y / CH, chunk = blockIdx.y - h * CH;
  int b = blockIdx.z, col0 = 0, ks = KSP;
  const int slot = blockIdx.z;
  if (items) {
    const int4 it = items[blockIdx.z];
    b = 0; col0 = it.x; Tq = it.y;
    ks = attn_ksplit_of(1, it.y, it.y, ks_force);
    if (ks == 1 || blockIdx.x * 64 >= it.y) return;      // an unsplit item was written normalised by the attention kernel itself
  }
  const int qc = min(q, Tq - 1);
  const float NEG = -__builtin_inff();
  float w[KSP], lv[KSP], M = NEG;
#pragma unroll
  for (int s = 0; s < KSP; ++s) {             // (loads are unconditional -- a split the item does not have reads a sibling's slot -- and selected)
    const float* ml = ML + ((((long)s * B + slot) * kHeads + h) * 2) * TqR;
    w[s] = ml[qc];
    lv[s] = ml[TqR + qc];
  }
  const long off = b * o_bs + (long)(h * DH + chunk * 16 + cg * 4) * ldo + col0 + qc;
  float pv[4][KSP];
#pragma unroll
  for (int ci = 0; ci < 4; ++ci)
#pragma unroll
    for (int s = 0; s < KSP; ++s) pv[ci][s] = P[s * p_split + off + (long)ci * ldo];
#pragma unroll
  for (int s = 0; s < KSP; ++s) {
    w[s] = s < ks ? w[s] : NEG;
    lv[s] = s < ks ? lv[s] : 0.f;
    M = fmaxf(M, w[s]);
  }
  const float M_use = M == NEG ? 0.f : M;
  float L = 0.f;
#pragma unroll
  for (int s = 0; s < KSP; ++s) {
    w[s] = __builtin_amdgcn_exp2f(w[s] - M_use);
    L += lv[s] * w[s];
  }
  const float rl = 1.f / L;
  if (q >= Tq) return;
#pragma unroll
  for (int ci = 0; ci < 4; ++ci) {
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < KSP; ++s) acc += (s < ks ? pv[ci][s] : 0.f) * w[s];
    O[off + (long)ci * ldo] = acc * rl;
  }
}

inline int& attn_ksplit_forced() {   // FDX_ATTN_KSPLIT=<n> (A/B); the ubench sets it directly
  static int v = [] { const char* e = getenv("FDX_ATTN_KSPLIT"); return e ? atoi(e) : 0; }();
  return v;
}
inline int attn_ksplit(int B, int Tq, int Tk) { return attn_ksplit_of(B, Tq, Tk, attn_ksplit_forced()); }
// floats of scratch the split path needs: partial O^T per split + (max, sum) per query
// (an exact-ragged row -- n_items > 0, B == 1 -- splits every item up to 8 ways, each item with its own (max, sum) slots)
inline size_t attn_part_floats(int B, int T, int D, int ld, int n_items = 0) { return (size_t)(n_items ? 8 : attn_ksplit(B, T, T)) * B * D * ld; }
inline size_t attn_ml_floats(int B, int T, int n_items = 0, int max_len = 0) {
  if (n_items) return (size_t)8 * n_items * kHeads * 2 * round_up(max_len, 128);
  return (size_t)attn_ksplit(B, T, T) * B * kHeads * 2 * round_up(T, 128);
}
struct AttnItems { const int4* dev = nullptr; const std::vector<int>* host = nullptr; int max_len = 0; };   // exact-ragged row layout (fdx_ctx::items)

hipError_t launch_attn_qs(int DH, AttnArgs a, int B, hipStream_t s, ProfEvents* prof, const AttnItems& items = AttnItems{}) {
  a.items = nullptr; a.ks_force = attn_ksplit_forced();
  int n_z = B, Tq_grid = a.Tq;
  if (items.dev) {                           // one row, grid.z = items; the grid is sized for the longest item and the finest split
    a.items = items.dev;
    n_z = (int)items.host->size() / 2;
    Tq_grid = items.max_len;
    a.B = n_z;
    a.ksplit = 1;
    for (int i = 0; i < n_z; ++i) a.ksplit = max(a.ksplit, attn_ksplit_of(1, (*items.host)[2 * i + 1], (*items.host)[2 * i + 1], a.ks_force));
    a.TqR = round_up(items.max_len, 128);
    a.p_split = a.o_bs;                      // the row is ONE batch item
  } else {
    a.B = B;
    a.ksplit = (a.P && a.ML) ? attn_ksplit(B, a.Tq, a.Tk) : 1;
    a.TqR = round_up(a.Tq, 128);
    a.p_split = (long)B * a.o_bs;
  }
  const int n_qb = (Tq_grid + 127) / 128;
  const dim3 grid(kHeads * n_qb * a.ksplit, 1, n_z), blk(256);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // fdx_prof_*: QK^T and PV of one attention launch = 2 * 2 * Tq * Tk * D flops per item
  if (prof) {
    prof->note(PROF_TD_ATTN, "k_attn_qs<%d> (v_mfma_f32_32x32x2_f32 fp32 flash attention: 128-query workgroups share K / V tiles through LDS, keys split %d ways; %ld workgroups%s)",
               DH, a.ksplit, (long)grid.x * n_z, a.ksplit > 1 ? " + k_attn_combine, not in the timed interval" : "");
    double fl = 0;
    if (items.dev) for (int i = 0; i < n_z; ++i) fl += 4.0 * (double)(*items.host)[2 * i + 1] * (*items.host)[2 * i + 1] * (double)(DH * kHeads);
    else fl = 4.0 * (double)a.Tq * a.Tk * (double)(DH * kHeads) * B;
    prof->take(PROF_TD_ATTN, fl, ev0, ev1);
  }
#define FDX_COMBINE_ARGS a.P, a.ML, a.O, a.o_bs, a.p_split, a.ldo, a.TqR, a.Tq, a.B, a.items, a.ks_force
#define FDX_ATTN(DH_)                                                                                   \
  {                                                                                                     \
    if (ev0) hipExtLaunchKernelGGL((k_attn_qs<DH_>), grid, blk, 0, s, ev0, ev1, 0, a);                   \
    else hipLaunchKernelGGL((k_attn_qs<DH_>), grid, blk, 0, s, a);                                       \
    const dim3 cgrid((Tq_grid + 63) / 64, kHeads * (DH_ / 16), n_z);                                     \
    switch (a.ksplit) {                                                                                 \
      case 2: hipLaunchKernelGGL((k_attn_combine<DH_, 2>), cgrid, blk, 0, s, FDX_COMBINE_ARGS); break;                  \
      case 3: hipLaunchKernelGGL((k_attn_combine<DH_, 3>), cgrid, blk, 0, s, FDX_COMBINE_ARGS); break;                  \
      case 4: hipLaunchKernelGGL((k_attn_combine<DH_, 4>), cgrid, blk, 0, s, FDX_COMBINE_ARGS); break;                  \
      case 5: hipLaunchKernelGGL((k_attn_combine<DH_, 5>), cgrid, blk, 0, s, FDX_COMBINE_ARGS); break;                  \
      case 6: hipLaunchKernelGGL((k_attn_combine<DH_, 6>), cgrid, blk, 0, s, FDX_COMBINE_ARGS); break;                  \
      case 7: hipLaunchKernelGGL((k_attn_combine<DH_, 7>), cgrid, blk, 0, s, FDX_COMBINE_ARGS); break;                  \
      case 8: hipLaunchKernelGGL((k_attn_combine<DH_, 8>), cgrid, blk, 0, s, FDX_COMBINE_ARGS); break;                  \
      default: break;                                                                                   \
    }                                                                                                   \
  }
  if (DH == 64) FDX_ATTN(64)
  else if (DH == 32) FDX_ATTN(32)
  else FDX_ATTN(16)
#undef FDX_ATTN
#undef FDX_COMBINE_ARGS
  return hipGetLastError();
}

hipError_t launch_attn(int DH, const AttnArgs& a, int B, hipStream_t s, ProfEvents* prof = nullptr, const AttnItems& items = AttnItems{}) {
  return launch_attn_qs(DH, a, B, s, prof, items);
}

// Scratch of one decoder layer (owned by the caller; padded rows [B][ch][ld], pointers past the left halo)
struct DecScratch { float* QKV; float* O; float* G; float* P = nullptr; float* ML = nullptr; };   // [3D], [D], [H]; attention split scratch (attn_part_floats / attn_ml_floats; null = never split)

// x = norm1(x + out_proj(attn(in_proj(x)))); x = norm2(x + out_proj(attn(q(x), K, V))); x = norm3(x + linear2(gelu(linear1(x)))), in place
// on X.  The cross-attention keys / values come from `KV` [B][2D][ld] (already projected with y.ca_kv: the caller decides whether
// that projection is per call or hoisted; kv_bs = floats between batch items of KV).  tgt_kpm / mem_kpm: [B][T] key-padding masks (1 = ignored) or null.
// `ca_bias` (optional): the cross-attention out-projection's bias as a per-item column of a table, bias(row, item) = ca_bias[row * ca_bias_ld +
// item * ca_bias_bs] -- the transformer denoiser folds the diffusion step's path through the value projection into it (tfdec.hip).
inline hipError_t run_declayer(const float* A, const TdLayer& y, int B, int T, int D, int H, int ld, float* X, const float* KV, long kv_bs,
                               const DecScratch& sc, const uint8_t* tgt_kpm, const uint8_t* mem_kpm, hipStream_t s, ProfEvents* prof = nullptr,
                               const float* ca_bias = nullptr, int ca_bias_ld = 1, int ca_bias_bs = 0, const AttnItems& items = AttnItems{}) {
  const long bsD = (long)D * ld, bsH = (long)H * ld;
  const int DH = D / kHeads;
  auto residual = [&](const PackedW& p, const float* in, long in_bs, const float* bias = nullptr, int b_ld = 1, int b_bs = 0) {   // X += W in + b
    EpiScaleRes e{};
    e.X = X; e.bs = bsD; e.ld = ld; e.bias = bias ? bias : A + p.b_off; e.bias_ld = b_ld; e.bias_bs = b_bs; e.gamma = nullptr; e.M = D;
    e.mask = nullptr; e.mask_ld = T;
    return gemm(A, p, B, T, in, in_bs, ld, e, s);
  };
  hipError_t e;
  AttnArgs at{};
  at.O = sc.O; at.o_bs = bsD; at.ldo = ld; at.Tq = T; at.Tk = T; at.scale = 1.f / sqrtf((float)DH);
  at.P = sc.P; at.ML = sc.ML;
  // ---- self-attention block
  if ((e = gemm(A, y.sa_in, B, T, X, bsD, ld, bias_epi(sc.QKV, 3 * bsD, ld, A + y.sa_in.b_off, 3 * D, ACT_NONE), s)) != hipSuccess) return e;
  at.Q = sc.QKV; at.q_bs = 3 * bsD; at.ldq = ld;
  at.K = sc.QKV + (size_t)D * ld; at.k_bs = 3 * bsD; at.ldk = ld;
  at.V = sc.QKV + (size_t)2 * D * ld; at.v_bs = 3 * bsD; at.ldv = ld;
  at.kmask = tgt_kpm;
  if ((e = launch_attn(DH, at, B, s, prof, items)) != hipSuccess) return e;
  if ((e = residual(y.sa_out, sc.O, bsD)) != hipSuccess) return e;
  launch_layernorm(X, bsD, ld, A + y.n1w, A + y.n1b, B, D, T, s);
  // ---- cross-attention block
  if ((e = gemm(A, y.ca_q, B, T, X, bsD, ld, bias_epi(sc.QKV, 3 * bsD, ld, A + y.ca_q.b_off, D, ACT_NONE), s)) != hipSuccess) return e;
  at.K = KV; at.k_bs = kv_bs; at.V = KV + (size_t)D * ld; at.v_bs = kv_bs;
  at.kmask = mem_kpm;
  if ((e = launch_attn(DH, at, B, s, prof, items)) != hipSuccess) return e;
  if ((e = residual(y.ca_out, sc.O, bsD, ca_bias, ca_bias_ld, ca_bias_bs)) != hipSuccess) return e;
  launch_layernorm(X, bsD, ld, A + y.n2w, A + y.n2b, B, D, T, s);
  // ---- feed-forward block
  if ((e = gemm(A, y.lin1, B, T, X, bsD, ld, bias_epi(sc.G, bsH, ld, A + y.lin1.b_off, H, ACT_GELU), s)) != hipSuccess) return e;
  if ((e = residual(y.lin2, sc.G, bsH)) != hipSuccess) return e;
  launch_layernorm(X, bsD, ld, A + y.n3w, A + y.n3b, B, D, T, s);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ the same layer without LayerNorm launches
// Round 6.  Post-norm makes every LayerNorm's output both the next sublayer's input AND its residual; rounds 4-5 ran k_td_layernorm three times
// per layer (36 launches of 5.8 us = 10 % of a transformer-denoiser call, none of it matrix work).  Here the residual stream X stays
// UN-normalised with the statistics of its 32-channel groups per frame in `st[cur]`:
//   consumers (self-attention in-projection, cross-attention query projection, linear1) read it through PRE_LNP with the pending norm's affine
//     part folded into their weights at pack time (pack_declayer);
//   the three residual GEMMs normalise the OLD value on the fly, add their result, and emit the new value with new statistics
//     (EpiResLN / PRE_RESLN, convgemm.hip.h) into the other statistics buffer (other row groups of the same columns still read the old one).
// `ln` carries the pending norm between sublayers and across layers; after the last layer the caller's next GEMM consumes it the same way.
struct LnStream {
  float* st[2] = {nullptr, nullptr};     // [B][T][2][16] each
  int cur = 0;
  const float* pw = nullptr; const float* pb = nullptr;   // pending LayerNorm of X (null: X is a plain tensor)
};

template <class Epi>
inline hipError_t gemm_ln(const float* A, const PackedW& p, size_t R_off, int B, int T, int D, const float* X, long x_bs, int ldx, const LnStream& ln,
                          const Epi& e, hipStream_t s) {
  if (!ln.pw) return gemm(A, p, B, T, X, x_bs, ldx, e, s);
  ConvGeom g{B, T, p.cin8, 1, 0, 0, p.n_mtiles};
  const float4* Wp = reinterpret_cast<const float4*>(A + p.w_off);
  if (p.RB == 1) return launch_convgemm<1, true, PRE_LNP, Epi>(g, Wp, X, x_bs, ldx, 1.f, e, s, nullptr, nullptr, ln.st[ln.cur], A + R_off, D / 32, 1e-5f);
  return launch_convgemm<2, true, PRE_LNP, Epi>(g, Wp, X, x_bs, ldx, 1.f, e, s, nullptr, nullptr, ln.st[ln.cur], A + R_off, D / 32, 1e-5f);
}

// linear1 (dim -> mlp_factor dim, GELU, pending norm folded in) on the shape-adaptive split-K 16x16x4 family (convgemm16s.hip.h PRE_LNP; second session of
// round 6, after ConvNext's pwconv1 -- the same [2048 x 512] GEMM): its weights in the NR = 4 / NR = 2 fragment orders, derived at attach (tfdec.hip)
struct Lin1On16 { const float* w4 = nullptr; const float* w2 = nullptr; int nr = 4, nm = 4; };

inline hipError_t run_declayer_ln(const float* A, const TdLayer& y, int B, int T, int D, int H, int ld, float* X, const float* KV, long kv_bs,
                                  const DecScratch& sc, const uint8_t* tgt_kpm, const uint8_t* mem_kpm, hipStream_t s, ProfEvents* prof,
                                  const float* ca_bias, int ca_bias_ld, int ca_bias_bs, const AttnItems& items, LnStream& ln,
                                  const Lin1On16* l16 = nullptr) {
  const long bsD = (long)D * ld, bsH = (long)H * ld;
  const int DH = D / kHeads;
  // X = LN_pending(X) + W in + b, un-normalised and centred, statistics to the other buffer; `nw / nb` = the norm that follows this sublayer
  auto residual = [&](const PackedW& p, const float* in, long in_bs, const float* nw, const float* nb, const float* bias = nullptr, int b_ld = 1,
                      int b_bs = 0) {
    EpiResLN e{};
    e.X = X; e.bs = bsD; e.ld = ld; e.bias = bias ? bias : A + p.b_off; e.bias_ld = b_ld; e.bias_bs = b_bs;
    e.lnw = ln.pw; e.lnb = ln.pb; e.st_out = ln.st[ln.cur ^ 1]; e.M = D; e.T = T;
    ConvGeom g{B, T, p.cin8, 1, 0, 0, p.n_mtiles};
    const hipError_t rc = launch_convgemm<1, true, PRE_RESLN, EpiResLN>(g, reinterpret_cast<const float4*>(A + p.w_off), in, in_bs, ld, 1.f, e, s, nullptr,
                                                                         nullptr, ln.pw ? ln.st[ln.cur] : nullptr, nullptr, D / 32, 1e-5f);
    ln.cur ^= 1; ln.pw = nw; ln.pb = nb;
    return rc;
  };
  hipError_t e;
  AttnArgs at{};
  at.O = sc.O; at.o_bs = bsD; at.ldo = ld; at.Tq = T; at.Tk = T; at.scale = 1.f / sqrtf((float)DH);
  at.P = sc.P; at.ML = sc.ML;
  // ---- self-attention block
  if ((e = gemm_ln(A, y.sa_in, y.sa_in_R, B, T, D, X, bsD, ld, ln, bias_epi(sc.QKV, 3 * bsD, ld, A + y.sa_in.b_off, 3 * D, ACT_NONE), s)) != hipSuccess) return e;
  at.Q = sc.QKV; at.q_bs = 3 * bsD; at.ldq = ld;
  at.K = sc.QKV + (size_t)D * ld; at.k_bs = 3 * bsD; at.ldk = ld;
  at.V = sc.QKV + (size_t)2 * D * ld; at.v_bs = 3 * bsD; at.ldv = ld;
  at.kmask = tgt_kpm;
  if ((e = launch_attn(DH, at, B, s, prof, items)) != hipSuccess) return e;
  if ((e = residual(y.sa_out, sc.O, bsD, A + y.n1w, A + y.n1b)) != hipSuccess) return e;
  // ---- cross-attention block
  if ((e = gemm_ln(A, y.ca_q, y.ca_q_R, B, T, D, X, bsD, ld, ln, bias_epi(sc.QKV, 3 * bsD, ld, A + y.ca_q.b_off, D, ACT_NONE), s)) != hipSuccess) return e;
  at.K = KV; at.k_bs = kv_bs; at.V = KV + (size_t)D * ld; at.v_bs = kv_bs;
  at.kmask = mem_kpm;
  if ((e = launch_attn(DH, at, B, s, prof, items)) != hipSuccess) return e;
  if ((e = residual(y.ca_out, sc.O, bsD, A + y.n2w, A + y.n2b, ca_bias, ca_bias_ld, ca_bias_bs)) != hipSuccess) return e;
  // ---- feed-forward block
  if (l16 && ln.pw) {
    const ConvGeom g4{B, T, y.lin1.cin8, 1, 0, 0, H / 64}, g2{B, T, y.lin1.cin8, 1, 0, 0, H / 32};
    e = hipErrorInvalidValue;
#define FDX_LIN1_SHAPE(NR_, NM_)                                                                                                            \
  if (l16->nr == NR_ && l16->nm == NM_) {                                                                                                  \
    const EpiBiasAct16S<NM_> ea{sc.G, bsH, ld, A + y.lin1.b_off, ACT_GELU};                                                                 \
    e = launch_convgemm16s<EpiBiasAct16S<NM_>, NR_, NM_, PRE_LNP>(NR_ == 4 ? g4 : g2, NR_ == 4 ? l16->w4 : l16->w2, X, bsD, ld, ea, s, nullptr, \
                                                                 nullptr, ln.st[ln.cur], A + y.lin1_R, D / 32, 1e-5f);                       \
  }
    FDX_LIN1_SHAPE(4, 4) FDX_LIN1_SHAPE(4, 5) FDX_LIN1_SHAPE(4, 6) FDX_LIN1_SHAPE(4, 7) FDX_LIN1_SHAPE(4, 8)
    FDX_LIN1_SHAPE(2, 4) FDX_LIN1_SHAPE(2, 5) FDX_LIN1_SHAPE(2, 6) FDX_LIN1_SHAPE(2, 7) FDX_LIN1_SHAPE(2, 8)
#undef FDX_LIN1_SHAPE
    if (e != hipSuccess) return e;
  } else if ((e = gemm_ln(A, y.lin1, y.lin1_R, B, T, D, X, bsD, ld, ln, bias_epi(sc.G, bsH, ld, A + y.lin1.b_off, H, ACT_GELU), s)) != hipSuccess) return e;
  if ((e = residual(y.lin2, sc.G, bsH, A + y.n3w, A + y.n3b)) != hipSuccess) return e;
  return hipGetLastError();
}

}  // namespace
}  // namespace fdx
