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
// One workgroup = 128 queries x 1 head (32 per wave), its 4 waves sharing every K / V tile through LDS; the KEY range is split over up to 8
// workgroups whose (max, sum, O) triples a small combine kernel folds in a fixed order (declayer.hip.h, k_attn_qs / k_attn_combine).
// Hoisted per utterance batch (fdx_tfdec_prepare): C0 = condition_projection(conditioner) + positional embedding, and every layer's
// cross-attention keys / values of C0.  The reference's memory is mask(C0 + step 1^T) (convnext.py:353,359-360), step = the diffusion-step
// MLP's output, constant over frames, and the projections are linear:
//     K = Wk C0 + bk + (Wk step) 1^T        a per-query constant on every score: softmax cancels it (round 5: dropped, not computed)
//     V = Wv C0 + bv + (Wv step) 1^T        softmax rows sum to 1: the attention output gains exactly Wv step per channel,
// which the out-projection turns into a per-(layer, step) BIAS, Wo (Wv step) + bo -- computed for all timesteps of a sampler run next to the
// step embeddings (fdx_td_embed).  Masked memory frames only ever meet the softmax as ignored keys (memory_key_padding_mask is the same
// cond_masks), so nothing else of the mask survives.  Round 4 rebuilt the memory and ran 12 [2D x D] projections per denoiser call.
#include "common.hip.h"
#include "elementwise.hip.h"
#include "declayer.hip.h"

using namespace fdx;

namespace {

struct TdLayout {
  PackedW in0, in2, emb1, emb3, cond0, cond2, out0, out2;
  size_t pos = 0;            // positional_embedding [n_positions][D], raw
  size_t scale_q = 0, scale_k = 0;
  std::vector<TdLayer> layers;
  bool folded = false;       // LayerNorms folded into their consumers (declayer.hip.h run_declayer_ln)
  size_t out0_R = 0;         // row sums of output_projection.0 with the last layer's norm3 folded in
  size_t total_floats = 0;
};

// FDX_TD_LNFOLD=0: the k_td_layernorm launches of rounds 4-5 (A/B).  Decides the arena layout: read once per process.
bool td_fold_ln() {
  static const bool v = [] { const char* e = getenv("FDX_TD_LNFOLD"); return !e || atoi(e) != 0; }();
  return v;
}

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
  l.in0 = plan32(cur, H, d.mel_channels);      // (32-row tiles: see plan_declayer)
  l.in2 = plan32(cur, D, H);
  l.emb1 = plan64(cur, H, D);
  l.emb3 = plan64(cur, D, H);
  l.cond0 = plan64(cur, H, d.condition_dim);
  l.cond2 = plan32(cur, D, H);
  l.folded = td_fold_ln();
  l.layers.assign(d.num_layers, TdLayer{});
  for (auto& y : l.layers) plan_declayer(cur, y, D, H, l.folded);
  l.out0 = plan32(cur, D, D);
  if (l.folded) { l.out0_R = cur; cur += (size_t)round_up(D, 64); }
  l.out2 = plan32(cur, d.mel_channels, D);     // (28 workgroups of 64 rows spend 16 k cycles in their K loop: 16 us for 0.1 GFLOP; 56 of 32 rows half that)
  l.total_floats = cur;
}


struct TdBufs {
  DevBuf X, QKV, KVh, O, G, Hin, H2, C0, condp, c1, cmask;   // KVh: [B][L][2D][ld] hoisted cross-attention keys / values
  DevBuf AP, AML;   // attention: partial O^T and (max, sum) of the key splits (declayer.hip.h k_attn_qs)
  DevBuf ST;              // folded LayerNorms: two buffers of per-frame group statistics [B][T][2][16] (run_declayer_ln)
  DevBuf E, Hm, S0, SV, CB;   // per sampler run: step embeddings [D][n]; Wv step [D][n] (scratch); cross-attention out-projection bias [L][D][n]
  int ldn = 0;
};

}  // namespace

struct fdx_td_state {
  bool ok = false;
  fdx_tfdec_desc d{};
  TdLayout l;
  const float* arena = nullptr;
  TdBufs b;
  // linear1 of every layer once more in the 16x16x4 fragment orders (declayer.hip.h Lin1On16), derived on the device at attach; FDX_TD_LIN1_16S=0: off
  DevBuf lin1_16;
  std::vector<size_t> lin1_off4, lin1_off2;
  bool lin1_16_ok = false;
  int lin1_nr = 4, lin1_nm = 4;
};
static bool td_lin1_16s() { static const bool v = [] { const char* e = getenv("FDX_TD_LIN1_16S"); return !e || atoi(e) != 0; }(); return v; }

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
  const float* n3w = nullptr; const float* n3b = nullptr;       // the previous layer's norm3: folded into the next consumer of the stream
  for (auto& y : l.layers) {
    const float* const* wl = w + k;
    k += pack_declayer(A, y, wl, D, H, n3w, n3b);
    n3w = wl[16]; n3b = wl[17];
  }
  pack_lin_ln(A, l.out0, l.out0_R, w[k], D, D, w[k + 1], l.folded ? n3w : nullptr, n3b); k += 2;
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
  S->lin1_16_ok = false;
  if (td_lin1_16s() && S->l.folded && !S->l.layers.empty() && S->l.layers[0].lin1.RB == 1 && (d->dim * d->mlp_factor) % 64 == 0) {
    FDX_HIP(h, hipSetDevice(h->device));     // one-off at model load: default stream, synchronous
    size_t tot = 0;
    for (const auto& y : S->l.layers) tot += 2 * packed_floats(y.lin1.n_mtiles, 1, y.lin1.cin8, 1);
    FDX_HIP(h, S->lin1_16.ensure(tot * sizeof(float), false, nullptr));
    S->lin1_off4.clear(); S->lin1_off2.clear();
    size_t c = 0;
    for (const auto& y : S->l.layers) {
      const PackedW& p = y.lin1;
      const size_t nf = packed_floats(p.n_mtiles, 1, p.cin8, 1);
      S->lin1_off4.push_back(c); S->lin1_off2.push_back(c + nf);
      hipLaunchKernelGGL(k_repack16_from32rb1<4>, dim3((unsigned)((nf / 4 + 255) / 256)), dim3(256), 0, nullptr, S->lin1_16.f() + c, S->arena + p.w_off, p.n_mtiles, p.cin8);
      hipLaunchKernelGGL(k_repack16_from32rb1<2>, dim3((unsigned)((nf / 2 + 255) / 256)), dim3(256), 0, nullptr, S->lin1_16.f() + c + nf, S->arena + p.w_off, p.n_mtiles, p.cin8);
      c += 2 * nf;
    }
    FDX_HIP(h, hipDeviceSynchronize());
    S->lin1_16_ok = true;
  }
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
  // exact-ragged row (fdx_sampler_set_items): positions restart at every item, attention stays inside it
  const bool ragged = h->n_items() > 0;
  if (ragged && (B != 1 || h->items_T != T))
    return fail(h, FDX_E_ARG, "fdx_tfdec_prepare: the item layout describes one row of %d frames, got a batch of %d x %d (clear it with fdx_sampler_set_items(.., 0, ..))",
                h->items_T, B, T);
  const int Tpos = ragged ? h->items_max_len : T;
  if (Tpos > d.n_positions) return fail(h, FDX_E_ARG, "fdx_tfdec_prepare: %d frames exceed the positional table (%d)", Tpos, d.n_positions);
  const int* pidx = ragged ? static_cast<const int*>(h->pidx_dev.p) : nullptr;
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
  const int L = d.num_layers;
  // (X is linear1's B operand on the 16x16x4 family: its tiles may read up to 127 columns past T in the last row -- values unused, memory owned)
  FDX_HIP(h, b.X.ensure(sz(D) + kTailPad * sizeof(float), geom, s)); FDX_HIP(h, b.QKV.ensure(sz(3 * D), geom, s)); FDX_HIP(h, b.KVh.ensure(sz(2 * D) * L, geom, s));
  if (S->lin1_16_ok) {   // tile shape of linear1 for this geometry (ConvNext's pwconv1 rule: convnext.hip)
    const int rows16 = H / 16;
    Shape16 sh{4, 4};
    const long wg44 = (long)(rows16 / 4) * B * ((T + 63) / 64);
    if (wg44 < 2 * 256) sh = pick_shape16(rows16, B, T, 12000.0 / (32.0 * ((D / 8 + 3) / 4)));
    static const int forced = [] { const char* e = getenv("FDX_TD_LIN1_SHAPE"); return e ? atoi(e) : -1; }();
    if (forced > 0) sh = Shape16{forced / 10, forced % 10};
    else if (sh.NR == 4 && (long)(rows16 / 2) * B * ((T + 16 * sh.NM - 1) / (16 * sh.NM)) <= 512) sh.NR = 2;
    if ((sh.NR != 2 && sh.NR != 4) || sh.NM < 4 || sh.NM > 8) sh = Shape16{4, 4};
    S->lin1_nr = sh.NR; S->lin1_nm = sh.NM;
  }
  FDX_HIP(h, b.O.ensure(sz(D), geom, s)); FDX_HIP(h, b.G.ensure(sz(H), geom, s)); FDX_HIP(h, b.Hin.ensure(sz(H), geom, s));
  FDX_HIP(h, b.H2.ensure(sz(D), geom, s)); FDX_HIP(h, b.C0.ensure(sz(D), geom, s));
  FDX_HIP(h, b.condp.ensure(sz(E), geom, s)); FDX_HIP(h, b.c1.ensure(sz(H), geom, s));
  FDX_HIP(h, b.AP.ensure(attn_part_floats(B, T, D, ld, h->n_items()) * sizeof(float), false, s));
  FDX_HIP(h, b.AML.ensure(attn_ml_floats(B, T, h->n_items(), h->items_max_len) * sizeof(float), false, s));
  if (l.folded) FDX_HIP(h, b.ST.ensure((size_t)2 * B * T * 32 * sizeof(float), geom, s));
  // C0 = condition_projection(conditioner) + positional_embedding[:T] * position_scale_key   (convnext.py:348,353-357)
  hipLaunchKernelGGL(k_copy_rows, ew_grid(T, B * E), dim3(kEwBlock), 0, s, b.condp.f() + kHalo, (long)E * ld, ld, cond, (long)E * T, T, E, T,
                     1.f, (const uint8_t*)nullptr);
  FDX_HIP(h, gemm(A, l.cond0, B, T, b.condp.f() + kHalo, (long)E * ld, ld,
                  bias_epi(b.c1.f() + kHalo, (long)H * ld, ld, A + l.cond0.b_off, H, ACT_GELU), s));
  FDX_HIP(h, gemm(A, l.cond2, B, T, b.c1.f() + kHalo, (long)H * ld, ld,
                  bias_epi(b.C0.f() + kHalo, (long)D * ld, ld, A + l.cond2.b_off, D, ACT_NONE), s));
  hipLaunchKernelGGL(k_td_addpos, ew_grid(T, B * D), dim3(kEwBlock), 0, s, b.C0.f() + kHalo, (long)D * ld, ld, A + l.pos, A + l.scale_k,
                     (const uint8_t*)nullptr, D, T, pidx);
  // every layer's cross-attention keys / values of C0 (the step's share never needs projecting per call: see the header)
  for (int i = 0; i < L; ++i) {
    const auto& y = l.layers[i];
    FDX_HIP(h, gemm(A, y.ca_kv, B, T, b.C0.f() + kHalo, (long)D * ld, ld,
                    bias_epi(b.KVh.f() + kHalo + (size_t)i * 2 * D * ld, (long)L * 2 * D * ld, ld, A + y.ca_kv.b_off, 2 * D, ACT_NONE), s));
  }
  h->cond_masked = cond_mask != nullptr;
  if (cond_mask) {   // private copy: the key-padding mask of the cross-attention (dropped for PLMS' one unmasked call)
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
  // cross-attention out-projection bias per (layer, timestep): Wo (Wv step) + bo.  Wv = rows [D, 2D) of the packed [2D x D] key / value
  // projection = its second half of 64-row tiles, addressed in place.
  const int L = S->d.num_layers;
  FDX_HIP(h, b.SV.ensure((size_t)D * ldn * 4, geom, s));
  FDX_HIP(h, b.CB.ensure((size_t)L * D * ldn * 4, geom, s));
  for (int i = 0; i < L; ++i) {
    const auto& y = l.layers[i];
    PackedW wv = y.ca_kv;
    wv.rows = D; wv.n_mtiles = D / 64;
    wv.w_off = y.ca_kv.w_off + packed_floats(D / 64, 2, y.ca_kv.cin8, 1);
    FDX_HIP(h, gemm(A, wv, 1, n, b.S0.f() + kHalo, 0, ldn, bias_epi(b.SV.f() + kHalo, 0, ldn, nullptr, D, ACT_NONE), s));
    FDX_HIP(h, gemm(A, y.ca_out, 1, n, b.SV.f() + kHalo, 0, ldn,
                    bias_epi(b.CB.f() + kHalo + (size_t)i * D * ldn, 0, ldn, A + y.ca_out.b_off, D, ACT_NONE), s));
  }
  return FDX_OK;
}

// ================================================================================================ forward
int fdx_td_forward_core(fdx_ctx* h, const float* xin, int col0, int sb_bs, const uint8_t* mask, float* eps_out, long o_bs, int ldo,
                        hipStream_t s, bool unmasked_cond, const EpiUniPC* fuse) {
  fdx_td_state* S = td(h);
  const auto& d = S->d;
  const auto& l = S->l;
  const float* A = S->arena;
  const int D = d.dim, H = D * d.mlp_factor, M = d.mel_channels;
  const int B = h->B, T = h->T, ld = h->ld;
  TdBufs& b = S->b;
  const long bsD = (long)D * ld, bsH = (long)H * ld;
  float* X = b.X.f() + kHalo; float* QKV = b.QKV.f() + kHalo; const float* KVh = b.KVh.f() + kHalo; float* O = b.O.f() + kHalo;
  float* G = b.G.f() + kHalo; float* Hin = b.Hin.f() + kHalo; float* H2 = b.H2.f() + kHalo;
  const int L = d.num_layers;
  const uint8_t* cmask = (h->cond_masked && !unmasked_cond) ? static_cast<const uint8_t*>(b.cmask.p) : nullptr;
  const dim3 blk(kEwBlock);
  // x = input_projection(x)^T + pos * scale_q, masked                                         (convnext.py:343-346,356-357)
  FDX_HIP(h, gemm(A, l.in0, B, T, xin, (long)M * ld, ld, bias_epi(Hin, bsH, ld, A + l.in0.b_off, H, ACT_GELU), s));
  FDX_HIP(h, gemm(A, l.in2, B, T, Hin, bsH, ld, bias_epi(X, bsD, ld, A + l.in2.b_off, D, ACT_NONE), s));
  const bool ragged = h->n_items() > 0 && B == 1 && h->items_T == T;
  AttnItems items;
  if (ragged) { items.dev = static_cast<const int4*>(h->items_dev.p); items.host = &h->items; items.max_len = h->items_max_len; }
  hipLaunchKernelGGL(k_td_addpos, ew_grid(T, B * D), blk, 0, s, X, bsD, ld, A + l.pos, A + l.scale_q, mask, D, T,
                     ragged ? static_cast<const int*>(h->pidx_dev.p) : (const int*)nullptr);
  // memory = mask(C0 + diffusion_step) (:353,359-360) is never formed: its keys / values are the hoisted ones of C0, the step's share is the
  // per-step bias of each layer's cross-attention out-projection (column col0 of CB; sb_bs = its stride between batch items)
  const DecScratch sc{QKV, O, G, b.AP.f() + kHalo, b.AML.f()};
  if (l.folded) {
    LnStream ln;
    ln.st[0] = b.ST.f(); ln.st[1] = b.ST.f() + (size_t)B * T * 32;
    for (int i = 0; i < L; ++i) {
      Lin1On16 l16;
      if (S->lin1_16_ok) { l16.w4 = S->lin1_16.f() + S->lin1_off4[i]; l16.w2 = S->lin1_16.f() + S->lin1_off2[i]; l16.nr = S->lin1_nr; l16.nm = S->lin1_nm; }
      FDX_HIP(h, run_declayer_ln(A, l.layers[i], B, T, D, H, ld, X, KVh + (size_t)i * 2 * D * ld, (long)L * 2 * bsD, sc, mask, cmask, s, &h->prof,
                                 b.CB.f() + kHalo + (size_t)i * D * b.ldn + col0, b.ldn, sb_bs, items, ln, S->lin1_16_ok ? &l16 : nullptr));
    }
    // output_projection.0 reads the stream through the last layer's norm3 like every other consumer
    FDX_HIP(h, gemm_ln(A, l.out0, l.out0_R, B, T, D, X, bsD, ld, ln, bias_epi(H2, bsD, ld, A + l.out0.b_off, D, ACT_GELU), s));
  } else {
    for (int i = 0; i < L; ++i)
      FDX_HIP(h, run_declayer(A, l.layers[i], B, T, D, H, ld, X, KVh + (size_t)i * 2 * D * ld, (long)L * 2 * bsD, sc, mask, cmask, s, &h->prof,
                              b.CB.f() + kHalo + (size_t)i * D * b.ldn + col0, b.ldn, sb_bs, items));
    FDX_HIP(h, gemm(A, l.out0, B, T, X, bsD, ld, bias_epi(H2, bsD, ld, A + l.out0.b_off, D, ACT_GELU), s));
  }
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
