// convgemm.hip.h -- the one MFMA kernel family every dense contraction on the hot path goes through.
//
//   Y[b][m][t] = epilogue( sum_{tap} sum_{c} W[m][c][tap] * act(X[b][c][t + shift0 + tap*dshift]) )
//
// i.e. a (dilated, "same"-padded) Conv1d expressed as `taps` shifted GEMMs; taps == 1 is a plain GEMM
// (1x1 convs, Linear layers, the DFT and the mel filterbank).  gfx950 only.
//
// Design (MI355X-first):
//  * f32-in / f32-acc MFMA `v_mfma_f32_32x32x2_f32`: exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD,
//    the same rate as the VALU peak but issued by ONE instruction per 64 cycles, leaving the VALU free.
//  * a wave owns a [32*RB rows] x [64 cols] output tile = RB x 2 accumulators (f32x16 each).
//  * operands are never staged through LDS: MFMA at fp32 is slow enough (64 cyc) that L2->register
//    streaming keeps up.  A (weights) is pre-packed on the host in exact fragment order, so a wave reads
//    it with one coalesced 1 KiB `global_load_dwordx4` per 4 MFMA k-steps.  B (activations, [C][T] with T
//    contiguous) is read as 128-byte row segments; the conv taps are just shifted re-reads of the same
//    rows (L1/L2 hits), the zero halo around every row makes "same" padding free.
//  * two work decompositions:
//      SPLITK = true : the 4 waves of a workgroup split the K loop (channels x taps) of ONE 64-col tile and
//                      reduce through LDS -- 4x more workgroups for the batch-1 denoiser (T = 861 columns
//                      only give 14 column tiles; with 16 row tiles that is 224 workgroups for 256 CUs).
//      SPLITK = false: each wave owns its own 64-col tile (workgroup = 256 cols), no LDS, no barrier --
//                      the vocoder regime (10^5 columns).
//  * XCD-aware block order: consecutive logical tiles (same weight rows) land on the same XCD's L2.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <cstdlib>

namespace fdx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHalo = 32;   // zero columns kept left of t=0 (and >= that many right of T) in every padded row
constexpr int kWave = 64;

struct ConvArgs {
  const float4* __restrict__ Wp;  // packed weights: [m_tile][it = cb*taps+tap][rb][lane] float4
  const float* __restrict__ X;    // points at (b=0, c=0, t=0) of the padded input (halo already skipped)
  long x_bstride;                 // floats between batch items
  int ldx;                        // floats between channels
  int n_it;                       // (Cin/8) * taps
  int taps, shift0, dshift;       // column shift of tap j = shift0 + j*dshift
  int T;                          // valid columns per item
  int tiles_per_item;             // ceil(T / cols_per_block)
  int n_tiles_n;                  // B * tiles_per_item
  int n_mtiles;
  int xcd_rect;                   // tile -> XCD map: 0 = row runs (batch 1); large grids: 1 = 2 row halves x 4 column quarters, 2 = 4 row quarters x 2 column halves
  float in_slope;                 // leaky-relu slope applied to the B operand (PRE == 1 instantiations)
  // PRE == PRE_LN (LayerNorm over the K = channel axis folded into the GEMM, convnext.hip).  The B operand holds
  // GROUP-centred values u - mean_g (groups of 32 channels); col_stats [item][T][2][16] = {mean_g[16], M2_g[16]} per frame;
  // ln_R [rows][16] = per-group row sums of the weights.  result = rstd[t] * (acc + sum_g R[row][g] (mean_g[t] - mean[t])).
  const float* __restrict__ col_stats;
  const float* __restrict__ ln_R;
  int n_groups;
  float ln_eps;
#ifdef FDX_KTRACE
  unsigned long long* trace;      // [block][wave][8] shader-clock stamps of this launch, or null (tools/ktrace.py)
#endif
};

// Kernel-argument passing: the fields every workgroup needs before it can issue its first operand load travel as the kernel's
// first 14 scalar dwords, which `-mllvm -amdgpu-kernarg-preload-count=14` (fish_diffusion_amd/_build.py) has the dispatcher place
// in SGPRs at wave launch; by-value structs are never preloaded, they are fetched with s_load from the kernarg segment -- a
// cold read from beyond the L2 on every launch of a replayed graph, i.e. a fabric round trip before the first address can be
// formed.  The rest (ConvArgsCold, the epilogue) is only needed behind the pipeline prologue.  For the same reason the kernels
// never read gridDim (a hidden kernel argument, i.e. another kernarg-segment load): the grid size is n_tiles_n * n_mtiles
// and tiles_per_item is recomputed from T.
struct ConvArgsCold {
  float in_slope;
  const float* __restrict__ col_stats;
  const float* __restrict__ ln_R;
  int n_groups;
  float ln_eps;
#ifdef FDX_KTRACE
  unsigned long long* trace;
#endif
};
#define FDX_CONV_HOT_PARAMS                                                                                                   \
  const float4 *__restrict__ h_Wp, const float *__restrict__ h_X, long h_xbs, int h_ldx, int h_n_it, int h_taps, int h_shift0, \
      int h_dshift, int h_T, int h_ntn, int h_nmt
// (the tile -> XCD map choice rides in the signs of the n_mtiles / n_tiles_n dwords: hot arguments, no extra kernarg load)
#define FDX_CONV_HOT_ARGS(a) (a).Wp, (a).X, (a).x_bstride, (a).ldx, (a).n_it, (a).taps, (a).shift0, (a).dshift, (a).T, \
  ((a).xcd_rect == 2 ? -(a).n_tiles_n : (a).n_tiles_n), ((a).xcd_rect ? -(a).n_mtiles : (a).n_mtiles)

// Tile -> XCD map of the 16x16x4 residual-block kernels (block b runs on XCD b % 8, the 8 L2s are private).
//   row runs (batch 1):  XCD x owns a contiguous run of G/8 logical tiles in row-major order = a couple of row tiles x all column tiles:
//     the weights are fetched once chip-wide, the (small) activation operand once per XCD.  Right while the activations fit an L2.
//   rectangles (large grids, `xcd_rect`):  XCD x owns row half (x & 1) x column quarter (x >> 1), walked row-fastest: the resident
//     workgroups of an XCD cover ALL its row tiles (their weights, 3.1 MB for the dilated conv, stay L2-resident by constant re-use)
//     and a few column tiles at a time (the activations stream through once).  Batch 16 x 10 s, dilated conv + gate: 625 MB of HBM
//     traffic per launch with row runs (every XCD streams all 28 MB of activations twice) against 119 MB algorithmic.
//   The host only sets `xcd_rect` when the row tiles divide by the row groups.  The column tiles need not divide (round 6: the exact-ragged
//   serving micro-batches are 93-104 column tiles wide and ran row runs, 252 MB per launch for 53 MB of work): every column group is
//   ceil(n / groups) tiles wide, the launch is PADDED to 8 equal rectangles (conv_rect_grid) and a workgroup whose tile falls off the end
//   returns at once -- at most (groups - 1) x (row tiles per group) idle workgroups, 16 of 1504 at 93 x 16.  Returns false for those.
__device__ __forceinline__ bool conv_tile_of_block(int n_tiles_n, int n_mt, int rect, int bid, int& mt, int& nt) {
  const int G = n_tiles_n * n_mt, xcd = bid & 7, slot = bid >> 3;     // G == gridDim.x (row runs), from preloaded arguments
  if (rect) {
    const int rs = rect == 2 ? 2 : 1;                                  // log2(row groups)
    const int MH = n_mt >> rs, QC = 8 >> rs, NQ = (n_tiles_n + QC - 1) / QC;
    const int ntl = slot / MH;
    mt = (xcd & ((1 << rs) - 1)) * MH + (slot - ntl * MH);
    nt = (xcd >> rs) * NQ + ntl;
    return nt < n_tiles_n;
  }
  const int q8 = G >> 3, r8 = G & 7;
  const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  mt = L / n_tiles_n;
  nt = L - mt * n_tiles_n;
  return true;
}
// workgroups to launch for a tile grid under map `rect` (0: exactly the tiles)
inline int conv_rect_grid(int n_tiles_n, int n_mt, int rect) {
  if (!rect) return n_tiles_n * n_mt;
  const int rs = rect == 2 ? 2 : 1, QC = 8 >> rs;
  return 8 * (n_mt >> rs) * ((n_tiles_n + QC - 1) / QC);
}
// Grids of at least this many workgroups take the rectangle map (FDX_XCD_RECT=<n>; 0: never).  1024 = four rounds of the chip: below that the
// activations of a launch fit the L2s and row runs fetch the weights once chip-wide (the batch-1 headline: 256 workgroups).
inline long xcd_rect_min_grid() {
  static const long v = [] { const char* e = getenv("FDX_XCD_RECT"); return e ? atol(e) : 1024L; }();
  return v;
}
// Row split by measurement (batch 16 x 10 s, HBM bytes per launch from rocprofv3 PMC passes, profiles/r03_ddpm1000_pmc_traffic*.json;
// row runs | 2 row halves | 4 row quarters): dilated conv + gate (3 taps: 6.3 MB of weights) 625 | 302 | 232 MB against 119 MB algorithmic --
// a half's 3.1 MB of weights do not stay resident in a 4 MB L2 next to the streaming operands, a quarter's 1.6 MB do; out-projection
// (1 tap: 2.1 MB of weights) 615 | 208 | 262 MB against 171 MB -- fewer, wider row groups halve the activation re-reads instead.
// FDX_XCD_RECT_ROWS=2|4 forces one split (A/B runs).
inline int use_xcd_rect(int n_tiles_n, int n_mt, int taps) {
  static const int rows = [] { const char* e = getenv("FDX_XCD_RECT_ROWS"); return e ? atoi(e) : 0; }();
  const long g = (long)n_tiles_n * n_mt, m = xcd_rect_min_grid();
  if (m <= 0 || g < m) return 0;
  const bool ok2 = (n_mt & 1) == 0 && n_tiles_n >= 4, ok4 = (n_mt & 3) == 0 && n_tiles_n >= 2;
  if (rows == 2) return ok2 ? 1 : 0;
  if (rows == 4) return ok4 ? 2 : 0;
  if (taps > 1) return ok4 ? 2 : (ok2 ? 1 : 0);
  return ok2 ? 1 : (ok4 ? 2 : 0);
}
#define FDX_CONV_ARGS_FROM_HOT(cold)                                                                                   \
  ConvArgs a;                                                                                                          \
  a.Wp = h_Wp; a.X = h_X; a.x_bstride = h_xbs; a.ldx = h_ldx; a.n_it = h_n_it; a.taps = h_taps; a.shift0 = h_shift0;   \
  a.dshift = h_dshift; a.T = h_T; a.n_tiles_n = h_ntn < 0 ? -h_ntn : h_ntn; a.n_mtiles = h_nmt < 0 ? -h_nmt : h_nmt; \
  a.xcd_rect = h_nmt < 0 ? (h_ntn < 0 ? 2 : 1) : 0; \
  conv_args_cold(a, cold)

#ifdef FDX_KTRACE
#define FDX_STAMP(k) do { if (a.trace && lane == 0) a.trace[((long)blockIdx.x * (blockDim.x >> 6) + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
// slot 7: the wave's lifetime on the constant 100 MHz real-time counter (s_memrealtime): shader cycles / real time = the clock the kernel ran at
#define FDX_STAMP_RT0() const unsigned long long fdx_rt0 = __builtin_amdgcn_s_memrealtime()
#define FDX_STAMP_RT1() do { if (a.trace && lane == 0) a.trace[((long)blockIdx.x * (blockDim.x >> 6) + wave) * 8 + 7] = __builtin_amdgcn_s_memrealtime() - fdx_rt0; } while (0)
// the last shader-clock stamp and the real-time reading taken BACK TO BACK, before either is stored (round 5: with FDX_STAMP(5); FDX_STAMP_RT1();
// the real-time window also held the wait for slot 5's own store to issue behind the epilogue's stores -- is that the "2.13 GHz"?)
#define FDX_STAMP_END() do { const unsigned long long fdx_c5 = __builtin_amdgcn_s_memtime(), fdx_r1 = __builtin_amdgcn_s_memrealtime();         \
    if (a.trace && lane == 0) { unsigned long long* fdx_t = a.trace + ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 8; fdx_t[5] = fdx_c5; fdx_t[7] = fdx_r1 - fdx_rt0; } } while (0)
struct TraceState { unsigned long long* buf = nullptr; int max_launches = 0, n = 0, blocks_cap = 0; };
inline TraceState g_trace;
#else
#define FDX_STAMP(k) do { } while (0)
#define FDX_STAMP_RT0() do { } while (0)
#define FDX_STAMP_RT1() do { } while (0)
#define FDX_STAMP_END() do { } while (0)
#endif

__device__ __forceinline__ void conv_args_cold(ConvArgs& a, const ConvArgsCold& c) {
  a.in_slope = c.in_slope; a.col_stats = c.col_stats; a.ln_R = c.ln_R; a.n_groups = c.n_groups; a.ln_eps = c.ln_eps;
#ifdef FDX_KTRACE
  a.trace = c.trace;
#endif
}
inline ConvArgsCold conv_cold_of(const ConvArgs& a) {
  ConvArgsCold c{};
  c.in_slope = a.in_slope; c.col_stats = a.col_stats; c.ln_R = a.ln_R; c.n_groups = a.n_groups; c.ln_eps = a.ln_eps;
#ifdef FDX_KTRACE
  c.trace = a.trace;
#endif
  return c;
}

// ------------------------------------------------------------------------------------------ epilogues
// Column mapping: lane li of MFMA column block nb (0/1) owns output column t0 + 2*li + nb, i.e. every lane owns an
// ADJACENT column pair (t, t+1), t even.  B operands are fetched as 8-byte pairs and every epilogue moves pairs.
//
// An epilogue has two halves so that its global reads can be issued BEFORE the K loop and land behind it:
//   Pre  load (b, row, t, two)               -- global loads only (conditioner slab, residual, skip, bias ...)
//   void store(b, row, t, two, v[, w], pre)  -- arithmetic + stores.  `two` = column t+1 is valid too.
// `kPaired` epilogues get the values of row and row + "pair distance" (accumulators rb=0 / rb=1): gate/filter, re/im.

typedef float f2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) f2u { float x, y; };   // dword-aligned pair: global ld/st at any even/odd float

// pair loads are unconditional: every source read this way is one of the library's own padded rows (>= kHalo floats of
// slack right of column T-1), never a caller tensor.
__device__ __forceinline__ f2 ld2(const float* p, bool /*two*/) {
  const f2u v = *reinterpret_cast<const f2u*>(p);
  return f2{v.x, v.y};
}
__device__ __forceinline__ void st2(float* p, f2 v, bool two) {
  if (two) { f2u u; u.x = v.x; u.y = v.y; *reinterpret_cast<f2u*>(p) = u; }
  else p[0] = v.x;
}

// Pair store into one of the library's own padded rows: unconditional 8-byte store; when column t+1 == T it receives 0,
// which is what the zero halo right of the row must hold anyway (keeps the epilogue free of per-lane branches).
__device__ __forceinline__ void st2p(float* p, f2 v, bool two) {
  f2u u; u.x = v.x; u.y = two ? v.y : 0.f;
  // Non-temporal store: an activation tile is consumed by the NEXT kernel, mostly from other XCDs, whose L2s are private --
  // keeping it dirty in this XCD's L2 only defers the write-back to the end-of-kernel release.  Measured: 88.70 -> 87.93 ms
  // per utterance (batch 1, 100 steps), vocoder unchanged.
  typedef float f2nt __attribute__((ext_vector_type(2), aligned(4)));
  __builtin_nontemporal_store(f2nt{u.x, u.y}, reinterpret_cast<f2nt*>(p));
}

// same store, cached normally: for tensors the SAME tile (hence the same XCD) reads back in the next layer (residual x, skip sum)
__device__ __forceinline__ void st2p_keep(float* p, f2 v, bool two) {
  f2u u; u.x = v.x; u.y = two ? v.y : 0.f;
  *reinterpret_cast<f2u*>(p) = u;
}

// x / c for a wave-uniform constant c with rc = RN(1/c): Markstein's sequence q = RN(x*rc); r = x - q*c (exact, fused);
// q' = RN(q + r*rc) returns the correctly rounded quotient (checked bit-identical to IEEE division on 8e7 random x for
// c = sqrt(2)) in 3 dependent VALU ops instead of the ~12-op v_div_scale / v_rcp / v_div_fixup expansion.
__device__ __forceinline__ float div_const(float x, float c, float rc) {
  const float q = x * rc;
  const float r = fmaf(-q, c, x);
  return fmaf(r, rc, q);
}
__device__ __forceinline__ f2 div_const(f2 x, float c, float rc) { return f2{div_const(x.x, c, rc), div_const(x.y, c, rc)}; }

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_MISH = 2, ACT_GELU = 3 };

// nn.GELU() (exact, erf form): 0.5 * x * (1 + erf(x / sqrt(2)))
// GELU (erf form, F.gelu's default): erf as one branch-free rational x P(x^2) / Q(x^2) on [-4, 4] (the fp32 fit Eigen / XLA use; 4.4e-7
// max abs error on erf, GELU within 1.4e-6 abs of an fp64 evaluation on [-6, 6] and exactly 0 / x beyond the clamp (tested to +-1e4) -- torch's own fp32 gelu sits at 1.2e-6).  The library
// erff() is two divergent branches per element, ~2.5x the vector instructions, and fp32 VALU work is additive to the MFMAs around it.
__device__ __forceinline__ float erf_f(float x) {
  x = fminf(fmaxf(x, -4.f), 4.f);
  const float x2 = x * x;
  float p = -2.72614225801306e-10f, q = -1.45660718464996e-05f;
  p = fmaf(p, x2, 2.77068142495902e-08f);   q = fmaf(q, x2, -2.13374055278905e-04f);
  p = fmaf(p, x2, -2.10102402082508e-06f);  q = fmaf(q, x2, -1.68282697438203e-03f);
  p = fmaf(p, x2, -5.69250639462346e-05f);  q = fmaf(q, x2, -7.37332916720468e-03f);
  p = fmaf(p, x2, -7.34990630326855e-04f);  q = fmaf(q, x2, -1.42647390514189e-02f);
  p = fmaf(p, x2, -2.95459980854025e-03f);
  p = fmaf(p, x2, -1.60960333262415e-02f);
  // exactly +-1 at the clamp (erf(4) rounds to 1 in fp32): the approximate reciprocal can leave the rational 1e-7 off 1 there, and gelu's
  // 0.5 x (1 + erf) would then grow like 1e-7 |x| in the negative tail instead of reaching 0
  const float r = x * p * __builtin_amdgcn_rcpf(q);
  return x2 >= 16.f ? copysignf(1.f, x) : r;
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erf_f(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float mish_f(float x) {
  // x * tanh(softplus(x)); F.softplus: beta=1, threshold=20 (wavenet.py:8-10)
  float sp = x > 20.f ? x : log1pf(expf(x));
  return x * tanhf(sp);
}

// An epilogue may offer `load_late` beside `load`: the variant for kernels that read the epilogue operands at store time (the no-split path),
// where wave-uniform `if (operand)` branches are cheaper than the branch-free form the split-K prefetch needs.
template <class E, class = void> struct epi_has_load_late : std::false_type {};
template <class E> struct epi_has_load_late<E, std::void_t<decltype(&E::load_late)>> : std::true_type {};
template <class E>
__device__ __forceinline__ typename E::Pre epi_load_late(const E& e, int b, int row, int t, bool two) {
  if constexpr (epi_has_load_late<E>::value) return e.load_late(b, row, t, two);
  else return e.load(b, row, t, two);
}

struct EpiBias {  // out = act(acc + bias[row]); masked columns -> 0; optional out2 = out + sb[row]
  static constexpr bool kPaired = false;
  float* out; long o_bs; int ldo;
  const float* bias;           // [M] or null
  int M, act;
  const uint8_t* mask; int mask_ld;   // [B][mask_ld] bytes, 1 = masked
  float* out2; long o2_bs; int ldo2;  // optional second output
  const float* sb; int sb_ld, sb_bs;  // out2 = v + sb[row*sb_ld + b*sb_bs]
  int tight;                          // 1: `out` is a caller tensor with no padding (row pitch may equal T); out2 is always padded
  int out2_zero_masked;               // 1: masked columns of out2 get 0 instead of sb (exact-ragged mode: nothing exists beyond an item's length)
  // Pre-loads are branch-free and touch no loaded value: every load is unconditional, from a pointer selected by the wave-uniform
  // conditions (an absent operand reads the output row instead: readable, ignored by store()), rows past M are clamped.
  // With `bool` members, an early return and `mask[..] != 0` in here hipcc drained all loads (vmcnt(0)) in front of
  // the K loop -- the operand stages queued behind the epilogue's loads included.
  // (The mask bytes are read in store(), behind `if (mask)`: pre-loaded, their zero-extension is a use in front of the K loop.)
  struct Pre { float bias, sb; };
  __device__ __forceinline__ Pre load(int b, int row, int /*t*/, bool /*two*/) const {
    Pre p;
    const int r = min(row, M - 1);
    const float* dummy = out + b * o_bs;
    p.bias = *(bias ? bias + r : dummy);
    p.sb = *(out2 ? sb + (long)r * sb_ld + b * sb_bs : dummy);
    return p;
  }
  __device__ __forceinline__ float act1(float v, float bv, bool masked) const {
    if (bias) v += bv;
    if (act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (act == ACT_MISH) v = mish_f(v);
    else if (act == ACT_GELU) v = gelu_f(v);
    return masked ? 0.f : v;
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 v, const Pre& p) const {
    if (row >= M) return;
    bool m0 = false, m1 = false;
    if (mask) { m0 = mask[(long)b * mask_ld + t] != 0; m1 = two && mask[(long)b * mask_ld + t + 1] != 0; }
    v.x = act1(v.x, p.bias, m0);
    v.y = act1(v.y, p.bias, m1);
    if (tight) st2(out + b * o_bs + (long)row * ldo + t, v, two);
    else st2p(out + b * o_bs + (long)row * ldo + t, v, two);
    if (out2) {
      f2 y{v.x + p.sb, v.y + p.sb};
      if (out2_zero_masked) { if (m0) y.x = 0.f; if (m1) y.y = 0.f; }
      st2p(out2 + b * o2_bs + (long)row * ldo2 + t, y, two);
    }
  }
};

struct EpiGate {  // wavenet.py:112-115: y = conv + conditioner (bias folded into P); z = sigmoid(gate)*tanh(filter)
  static constexpr bool kPaired = true;
  float* out; long o_bs; int ldo;
  const float* P; long p_bs; int ldp;  // [B][2C][ldp]: conditioner slab (+ conv bias + conditioner bias)
  int C;
  struct Pre { f2 pg, pf; };
  __device__ __forceinline__ Pre load(int b, int row, int t, bool two) const {
    Pre p{f2{0.f, 0.f}, f2{0.f, 0.f}};
    if (row >= C) return p;
    const float* q = P + b * p_bs + t;
    p.pg = ld2(q + (long)row * ldp, two);
    p.pf = ld2(q + (long)(row + C) * ldp, two);
    return p;
  }
  // sigmoid(g) * tanh(f) on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each):
  //   sigmoid(g) = 1 / (1 + e^-g),  tanh(f) = 1 - 2 / (e^2f + 1)  (saturates cleanly: e^2f = inf -> 1, 0 -> -1).
  // Absolute error ~1e-7 on a value in [-1, 1] -- fp32 round-off class, two orders below what the reference's own
  // CPU backends differ by; the end-to-end parity tests (1e-3 rel on mel after 100 steps) hold it.
  // Near 0 the exp form of tanh loses RELATIVE accuracy (1 - (1 - f)), so |f| < 0.15 takes the odd Taylor polynomial
  // f (1 - f^2/3 + 2f^4/15 - 17f^6/315)  (next term 62/2835 f^8 < 6e-9 there); both are evaluated, one is selected.
  __device__ __forceinline__ static float gate1(float g, float f) {
    const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-g));
    const float th_e = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * f) + 1.f);
    const float f2_ = f * f;
    const float th_p = f * (1.f + f2_ * (-0.33333334f + f2_ * (0.13333334f + f2_ * -0.053968254f)));
    return sg * (fabsf(f) < 0.15f ? th_p : th_e);
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 g, f2 f, const Pre& p) const {
    if (row >= C) return;
    g += p.pg; f += p.pf;
    st2p(out + b * o_bs + (long)row * ldo + t, f2{gate1(g.x, f.x), gate1(g.y, f.y)}, two);
  }
};

struct EpiResSkip {  // wavenet.py:117-120 + the skip sum of :228
  static constexpr bool kPaired = false;
  float* X; float* Y; float* SK; long bs; int ld;   // all [B][C][ld]; Y may be null (last layer)
  const float* bias;                                  // [2C]
  const float* sb; int sb_ld, sb_bs;                  // next layer's diffusion projection, [C][sb_ld]
  int C, skip_mode;                                   // 0 first (=), 1 middle (+=), 2 last ((+=)/sqrt(L)); 3 = first and last
  float inv_div, r_inv_div;                           // sqrt(n_layers) and RN(1/sqrt(n_layers))
  struct Pre { f2 old; float bias, sb; };
  // A 32-row accumulator block lies entirely on one side of C (C % 32 == 0), so "residual or skip half?" is decided
  // on a wave-uniform value (scalar branch) instead of per lane.
  __device__ __forceinline__ bool is_res(int row) const { return __builtin_amdgcn_readfirstlane(row) < C; }
  __device__ __forceinline__ Pre load(int b, int row, int t, bool two) const {
    Pre p{f2{0.f, 0.f}, 0.f, 0.f};
    p.bias = bias[row];
    if (is_res(row)) {
      p.old = ld2(X + b * bs + (long)row * ld + t, two);
      if (Y) p.sb = sb[(long)row * sb_ld + b * sb_bs];
    } else if (skip_mode == 1 || skip_mode == 2) {
      p.old = ld2(SK + b * bs + (long)(row - C) * ld + t, two);
    }
    return p;
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 v, const Pre& p) const {
    v += p.bias;
    if (is_res(row)) {
      const long o = b * bs + (long)row * ld + t;
      const f2 xn = div_const(p.old + v, 1.41421356237309504880f, 0.70710678118654752440f);
      st2p_keep(X + o, xn, two);
      if (Y) st2p(Y + o, xn + p.sb, two);
    } else {
      const long o = b * bs + (long)(row - C) * ld + t;
      f2 s = v;
      if (skip_mode == 1 || skip_mode == 2) s = p.old + v;
      if (skip_mode >= 2) s = div_const(s, inv_div, r_inv_div);
      st2p_keep(SK + o, s, two);
    }
  }
};

// ------------------------------------------------------------------------------------------ bf16 storage mode (opt-in)
// Activations of the two residual-block GEMMs in "C8-blocked" bf16: element (channel c, column t) of an item lives at
// ((c >> 3) * ld + t) * 8 + (c & 7), i.e. one 16-byte group holds 8 consecutive channels of one column -- exactly the B operand
// of v_mfma_f32_32x32x16_bf16 (lane = column, 8 consecutive k).  Accumulation, gates, residual stream and skip sum stay fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void stb2(__bf16* base, int ld, int c, int t, f2 v, bool two) {
  __bf16* p = base + ((long)(c >> 3) * ld + t) * 8 + (c & 7);
  p[0] = (__bf16)v.x;
  p[8] = (__bf16)(two ? v.y : 0.f);      // next column, same channel (column T of an odd-length row gets its zero)
}

// four consecutive channels c0 .. c0+3 (c0 % 4 == 0) of the column pair (t, t+1): one 8-byte store per column -- the four
// accumulator rows r = 4q .. 4q+3 a lane owns are exactly such a quad (acc_row), so a wave's split-K sites group naturally
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stb2x4(__bf16* base, int ld, int c0, int t, const f2 (&v)[4], bool two) {
  __bf16* p = base + ((long)(c0 >> 3) * ld + t) * 8 + (c0 & 7);
  *reinterpret_cast<bf16x4*>(p) = bf16x4{(__bf16)v[0].x, (__bf16)v[1].x, (__bf16)v[2].x, (__bf16)v[3].x};
  *reinterpret_cast<bf16x4*>(p + 8) = two ? bf16x4{(__bf16)v[0].y, (__bf16)v[1].y, (__bf16)v[2].y, (__bf16)v[3].y}
                                           : bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
}

struct EpiGateB : EpiGate {  // as EpiGate, the gated output goes to the blocked bf16 operand of the out-projection
  static constexpr bool kQuad = true;   // the kernel hands over a wave's four sites at once (store4)
  __bf16* outb; long ob_bs;   // [B][C/8][ld][8], item stride in bf16 elements
  __device__ __forceinline__ f2 value(f2 g, f2 f, const Pre& p) const {
    g += p.pg; f += p.pf;
    return f2{gate1(g.x, f.x), gate1(g.y, f.y)};
  }
  __device__ __forceinline__ void store4(int b, int row0, int t, bool two, const f2 (&v)[4]) const {
    stb2x4(outb + b * ob_bs, ldo, row0, t, v, two);
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 g, f2 f, const Pre& p) const {   // per-site form
    if (row < C) stb2(outb + b * ob_bs, ldo, row, t, value(g, f, p), two);
  }
};

struct EpiResSkipB : EpiResSkip {  // as EpiResSkip; the next conv's input Y = X + step goes out as blocked bf16
  __bf16* Yb; long yb_bs;
  __device__ __forceinline__ Pre load(int b, int row, int t, bool two) const {
    Pre p{f2{0.f, 0.f}, 0.f, 0.f};
    p.bias = bias[row];
    if (is_res(row)) {
      p.old = ld2(X + b * bs + (long)row * ld + t, two);
      if (Yb) p.sb = sb[(long)row * sb_ld + b * sb_bs];
    } else if (skip_mode == 1 || skip_mode == 2) {
      p.old = ld2(SK + b * bs + (long)(row - C) * ld + t, two);
    }
    return p;
  }
  static constexpr bool kQuadY = true;   // fp32 stores per site; the bf16 operand of the next conv in quads (store_y4)
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 v, const Pre& p, f2& yv) const {
    v += p.bias;
    if (is_res(row)) {
      const long o = b * bs + (long)row * ld + t;
      const f2 xn = div_const(p.old + v, 1.41421356237309504880f, 0.70710678118654752440f);
      st2p_keep(X + o, xn, two);
      yv = xn + p.sb;
    } else {
      const long o = b * bs + (long)(row - C) * ld + t;
      f2 s = v;
      if (skip_mode == 1 || skip_mode == 2) s = p.old + v;
      if (skip_mode >= 2) s = div_const(s, inv_div, r_inv_div);
      st2p_keep(SK + o, s, two);
    }
  }
  __device__ __forceinline__ void store_y4(int b, int row0, int t, bool two, const f2 (&v)[4]) const {
    if (Yb && is_res(row0)) stb2x4(Yb + b * yb_bs, ld, row0, t, v, two);
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 v, const Pre& p) const {   // per-site form (non-split tiles)
    f2 yv{0.f, 0.f};
    store(b, row, t, two, v, p, yv);
    if (Yb && is_res(row)) stb2(Yb + b * yb_bs, ld, row, t, yv, two);
  }
};

// The denoiser's last projection (wavenet.py:231-234: eps = output_projection(h) + bias, masked) with the UniPC corrector -- and the NEXT step's
// predictor -- applied to eps in the epilogue instead of by a separate elementwise launch (k_unipc_post / k_unipc_post_pre, elementwise.hip.h):
//   model_t = (x_t - sigma eps) / alpha;  x = x_base - aB (rho0 D1 + rho1 (model_t - m0))                      uni_pc.py:673-680
//   [next]  x_base' = c_x' x - c_m' model_t;  x_t' = x_base' - aB' (0.5 (m0 - model_t) / rk')                  uni_pc.py:664-671, 797-804
// Same expressions in the same order as those kernels (every product / quotient / sum its own rounding: -ffp-contract=off), on the value
// the EpiBias epilogue would have stored -- the sampler's state is bit-identical to the unfused sequence.  eps itself is not stored.
struct EpiUniPC {
  static constexpr bool kPaired = false;
  float* x; float* mt; float* xbase; float* xt; const float* m0; const float* m1; long bs; int ld;   // sampler state, padded [B][M][ld]
  const float* bias; int M;
  const uint8_t* mask; int mask_ld;
  float sigma, alpha, aB, rk, rho0, rho1; int order;
  float n_cx, n_cm, n_aB, n_rk; int n_order;     // n_order = 0: last step, no predictor behind the corrector
  struct Pre { f2 xt, xb, m0, m1; float bias; };
  __device__ __forceinline__ Pre load(int b, int row, int t, bool two) const {
    Pre p;
    const int r = min(row, M - 1);
    const long o = b * bs + (long)r * ld + t;
    p.xt = ld2(xt + o, two); p.xb = ld2(xbase + o, two); p.m0 = ld2(m0 + o, two); p.m1 = ld2(m1 + o, two);
    p.bias = bias[r];
    return p;
  }
  __device__ __forceinline__ void one(float eps, float xtv, float xbv, float m0v, float m1v, float& xn, float& mtv, float& xb2, float& xt2) const {
    mtv = (xtv - sigma * eps) / alpha;
    const float D1t = mtv - m0v;
    float corr;
    if (order == 2) {
      const float D1 = (m1v - m0v) / rk;
      corr = rho0 * D1 + rho1 * D1t;
    } else {
      corr = rho1 * D1t;
    }
    xn = xbv - aB * corr;
    xb2 = n_cx * xn - n_cm * mtv;
    xt2 = xb2;
    if (n_order == 2) {
      const float D1n = (m0v - mtv) / n_rk;
      xt2 = xb2 - n_aB * (0.5f * D1n);
    }
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 v, const Pre& p) const {
    if (row >= M) return;
    v += p.bias;
    if (mask) {
      if (mask[(long)b * mask_ld + t] != 0) v.x = 0.f;
      if (two && mask[(long)b * mask_ld + t + 1] != 0) v.y = 0.f;
    }
    float r0[4], r1[4];
    one(v.x, p.xt.x, p.xb.x, p.m0.x, p.m1.x, r0[0], r0[1], r0[2], r0[3]);
    one(v.y, p.xt.y, p.xb.y, p.m0.y, p.m1.y, r1[0], r1[1], r1[2], r1[3]);
    const f2 xn{r0[0], r1[0]}, mtv{r0[1], r1[1]}, xb2{r0[2], r1[2]}, xt2{r0[3], r1[3]};
    const long o = b * bs + (long)row * ld + t;
    st2(mt + o, mtv, two);
    st2(x + o, xn, two);
    if (n_order != 0) { st2(xbase + o, xb2, two); st2(xt + o, xt2, two); }
  }
};

struct EpiScaleRes {  // convnext.py:84-92: x = residual + gamma * (pwconv2(.) + bias); masked_fill(x_masks)
  static constexpr bool kPaired = false;
  float* X; long bs; int ld;                 // residual in, result out (in place), padded rows
  const float* bias; const float* gamma; int M;
  const uint8_t* mask; int mask_ld;
  int bias_ld = 1, bias_bs = 0;              // bias of (row, item) at bias[row * bias_ld + item * bias_bs]: a per-item column of a [M][n] table
  struct Pre { f2 old; float bias, gamma; };   // (branch-free, mask read in store(): see EpiBias::load)
  __device__ __forceinline__ Pre load(int b, int row, int t, bool two) const {
    Pre p;
    const int r = min(row, M - 1);
    p.old = ld2(X + b * bs + (long)r * ld + t, two);
    p.bias = bias[(long)r * bias_ld + b * bias_bs];
    p.gamma = *(gamma ? gamma + r : bias + r);                  // gamma == null: plain residual add x + (v + bias), see store()
    return p;
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 v, const Pre& p) const {
    if (row >= M) return;
    v = p.old + (gamma ? p.gamma : 1.f) * (v + p.bias);
    if (mask) {
      if (mask[(long)b * mask_ld + t] != 0) v.x = 0.f;
      if (two && mask[(long)b * mask_ld + t + 1] != 0) v.y = 0.f;
    }
    st2p_keep(X + b * bs + (long)row * ld + t, v, two);
  }
};

// Post-norm residual of nn.TransformerDecoderLayer with the LayerNorms FOLDED AWAY (round 6; tfdec.hip, declayer.hip.h run_declayer_ln):
//   r' = LN_pending(r) + (W in + bias)                       x = norm(x + sublayer(x)), one sublayer later
// The residual stream X is kept UN-normalised, with the means and centred sums of squares of its 32-row groups per frame in ST [item][T][2][16]
// (what PRE_LNP consumers combine into mean / rstd).  `old` is normalised on the fly from its own statistics (-mean and rstd arrive from the
// kernel's column-statistics stage, PRE_RESLN), the new value's group statistics are formed across the workgroup's four waves and written
// next to it.  lnw == null: `old` carries no pending norm (the first sublayer of the first layer).
// Only for convgemm_kernel<1, true, PRE_RESLN, EpiResLN>: the kernel calls value() and stores X / ST itself.
struct EpiResLN {
  static constexpr bool kPaired = false;
  float* X; long bs; int ld;
  const float* bias; int bias_ld, bias_bs;     // bias of (row, item) at bias[row * bias_ld + item * bias_bs]
  const float* lnw; const float* lnb;          // the pending LayerNorm's affine parameters, or null
  float* st_out;                               // [item][T][2][16]; never the buffer the kernel's col_stats point at (other row groups still read it)
  int M, T;
  struct Pre { f2 old; float bias, w, b; };
  __device__ __forceinline__ Pre load(int b, int row, int t, bool two) const {
    Pre p;
    const int r = min(row, M - 1);
    p.old = ld2(X + b * bs + (long)r * ld + t, two);
    p.bias = bias[(long)r * bias_ld + b * bias_bs];
    p.w = *(lnw ? lnw + r : bias + (long)r * bias_ld);
    p.b = *(lnw ? lnb + r : bias + (long)r * bias_ld);
    return p;
  }
  __device__ __forceinline__ f2 value(f2 v, const Pre& p, f2 neg_mean, f2 rstd) const {
    const f2 o = lnw ? ((p.old + neg_mean) * rstd) * p.w + p.b : p.old;
    return o + (v + p.bias);
  }
  __device__ __forceinline__ void store(int, int, int, bool, f2, const Pre&) const {}   // (unused: see above)
};

struct EpiResblock {  // models.py:103-110 conv2: x = xt + x; plus the MRF mean of :426-432
  static constexpr bool kPaired = false;
  float* out; const float* resid; long bs; int ld;
  const float* bias; int M;
  int mode;     // 0: out = v    1: out += v    2: out = (out + v) / div
  float div;
  // leaky-relu slope applied to the residual as it is added (1 = the plain `xt + x` of ResBlock1).  ResBlock2's `F.leaky_relu(x, .., inplace=True)`
  // (models.py:152) rewrites x before `xt + x` (:154): its residual is the ACTIVATED x.  x * 1.f is exact, so one code path serves both.
  float rslope = 1.f;
  struct Pre { f2 res, old; float bias; };
  // Branch-free (see EpiBias::load): rows past M are clamped, an absent operand reads a pair of the bias vector instead (cached, ignored
  // by store()).  With `if (resid) .. if (mode != 0) ..` here hipcc put an s_waitcnt vmcnt(0) behind each conditional load: the split-K
  // kernel's eight epilogue sites became eight serial round trips in front of the K loop on every launch that accumulates (mode != 0).
  __device__ __forceinline__ Pre load(int b, int row, int t, bool two) const {
    Pre p;
    const int r = min(row, M - 1);
    const long o = b * bs + (long)r * ld + t;
    const float* dummy = bias + (r & ~1);
    p.bias = bias[r];
    p.res = ld2(resid ? resid + o : dummy, two);
    p.old = ld2(mode != 0 ? out + o : dummy, two);
    return p;
  }
  __device__ __forceinline__ Pre load_late(int b, int row, int t, bool two) const {
    Pre p{f2{0.f, 0.f}, f2{0.f, 0.f}, 0.f};
    if (row >= M) return p;
    const long o = b * bs + (long)row * ld + t;
    p.bias = bias[row];
    if (resid) p.res = ld2(resid + o, two);
    if (mode != 0) p.old = ld2(out + o, two);
    return p;
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 v, const Pre& p) const {
    if (row >= M) return;
    v += p.bias;
    if (resid) v += f2{p.res.x > 0.f ? p.res.x : p.res.x * rslope, p.res.y > 0.f ? p.res.y : p.res.y * rslope};
    if (mode == 1) v = p.old + v;
    else if (mode == 2) v = (p.old + v) / div;
    st2p(out + b * bs + (long)row * ld + t, v, two);
  }
};

struct EpiUps {  // polyphase ConvTranspose1d (models.py:421): logical row = phase*Cout + co
  static constexpr bool kPaired = false;
  float* out; long o_bs; int ldo;
  const float* bias; int Cout, stride, Lout;
  struct Pre { float bias; };
  __device__ __forceinline__ Pre load(int b, int row, int t, bool two) const {
    const int ph = row / Cout;
    return Pre{ph < stride ? bias[row - ph * Cout] : 0.f};
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 v, const Pre& p) const {
    const int ph = row / Cout;
    if (ph >= stride) return;
    const int co = row - ph * Cout;
    float* o = out + b * o_bs + (long)co * ldo;
    const int n = t * stride + ph;
    if (n < Lout) o[n] = v.x + p.bias;
    if (two && n + stride < Lout) o[n + stride] = v.y + p.bias;
  }
};

struct EpiMag {  // pitch_adjustable_mel.py:83-92: sqrt(re^2 + im^2 + 1e-9) [* win_size / win_new]
  static constexpr bool kPaired = true;
  float* out; long o_bs; int ldo;
  int n_bins, n_rows;   // rows in [n_bins, n_rows) are written as 0 (zero-padded bins)
  float mul, div;       // 0 => no rescale
  struct Pre {};
  __device__ __forceinline__ Pre load(int, int, int, bool) const { return Pre{}; }
  __device__ __forceinline__ float mag1(float re, float im) const {
    float v = sqrtf(re * re + im * im + 1e-9f);
    if (mul != 0.f) v = v * mul / div;
    return v;
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 re, f2 im, const Pre&) const {
    if (row >= n_rows) return;
    f2 v{0.f, 0.f};
    if (row < n_bins) v = f2{mag1(re.x, im.x), mag1(re.y, im.y)};
    st2p(out + b * o_bs + (long)row * ldo + t, v, two);
  }
};

struct EpiLogMel {  // audio.py:11-18 + nsf_hifigan.py:104-105
  static constexpr bool kPaired = false;
  float* out; long o_bs; int ldo; int M, log_mode;
  struct Pre {};
  __device__ __forceinline__ Pre load(int, int, int, bool) const { return Pre{}; }
  __device__ __forceinline__ float log1(float v) const {
    if (log_mode != 0) {
      v = logf(fmaxf(v, 1e-5f));
      if (log_mode == 2) v = 0.434294f * v;
    }
    return v;
  }
  __device__ __forceinline__ void store(int b, int row, int t, bool two, f2 v, const Pre&) const {
    if (row >= M) return;
    st2(out + b * o_bs + (long)row * ldo + t, f2{log1(v.x), log1(v.y)}, two);
  }
};

// ------------------------------------------------------------------------------------------ kernel
// Accumulator element r of a 32x32 tile sits at row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31.
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

constexpr int OPK_F32 = 0, OPK_BF16 = 1;   // operand kind: fp32 rows (32x32x2 MFMA) / C8-blocked bf16 (32x32x16 MFMA)
constexpr int PRE_NONE = 0, PRE_LRELU = 1, PRE_LN = 2, PRE_RESLN = 3, PRE_LNP = 4;   // operand / result transforms (a bool converts: false/true = none/lrelu)
// PRE_LNP ("plain"): LayerNorm over K folded into the GEMM with the B operand left as it is: result = rstd[t] (acc - mean[t] rowsum[row]);
//   `ln_R` = [rows] row sums of the (affine-folded) weights.  One multiply-subtract per output instead of PRE_LN's 16-term group correction (which cost
//   the transformer's consumers 2-3 us per launch, round 6); exact to rounding while |mean| is not orders of magnitude above the deviation -- a
//   post-norm transformer's residual stream (LayerNorm output + sublayer) and, since the same round, ConvNext's depthwise-conv output; measured:
//   error ~ 5e-7 (1 + |mean| / sigma) of the normalised product (profiles/NOTES.md round 6).  FDX_CN_LNP=0 / FDX_TD_LNFOLD=0 keep the exact forms.
// PRE_RESLN (EpiResLN only): the column statistics belong to the epilogue's RESIDUAL operand, not to the GEMM's B operand -- same combine stage,
//   no row-sum correction; the epilogue emits the new value with its own group statistics.

// (PRE_LN keeps its statistics in registers across the K loop: the second launch-bounds argument holds that instantiation to
// the 256 VGPRs that let two workgroups share a CU, like every other instantiation already does unprompted.)
// (Measured and removed, round 4 -- see profiles/NOTES.md: 8 K-splitting waves per workgroup (2 % slower), 128-row MT = 2 tiles (+1 % at batch 2,
// -10 % at batch 8), a 2-D tile -> XCD map and a late epilogue prefetch for this family.)
// (Measured and removed, round 5 -- profiles/r05_mt2_and_splitk_rect_ab.txt: workgroups owning TWO packed m-tiles (128 x 64 tiles, 12 instead of 16
// B/clk/CU of operands, 224 instead of 448 workgroups for a 2048-row GEMM over 861 columns; bit-identical): ConvNext pwconv1 26.2 -> 32.6 us, the
// transformer's linear1 likewise -- one workgroup per CU loses the overlap two co-resident 64 x 64 workgroups give each other's fills and epilogues.)
template <int RB, bool SPLITK, int PRE, class Epi, int OPK = OPK_F32>
__global__ __launch_bounds__(256, PRE == PRE_LN ? 2 : 1) void convgemm_kernel(FDX_CONV_HOT_PARAMS, ConvArgsCold cold, Epi epi) {
  FDX_CONV_ARGS_FROM_HOT(cold);
  a.tiles_per_item = (a.T + (SPLITK ? 63 : 255)) / (SPLITK ? 64 : 256);
  constexpr int NW = 4;                       // waves per workgroup
  static_assert(!Epi::kPaired || RB == 2, "paired epilogues need both row blocks");
  constexpr int NB = 2;                       // two 32-column MFMA blocks per wave tile (interleaved columns)
  constexpr int RBX = RB;                     // 32-row accumulator blocks per wave
  constexpr int V = RBX * NB;                 // accumulator values per (lane, accumulator row r)
  constexpr int ROWS = Epi::kPaired ? 32 : 32 * RB;   // logical rows per packed m-tile
  __shared__ float red[SPLITK ? NW * V * 16 * kWave : 1];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, li = lane & 31;
  FDX_STAMP(0);

  // ---- XCD-aware logical tile id (block b runs on XCD b % 8): row runs, or -- when the activation operand outweighs the weights (round 5:
  // ConvNext pwconv2 at batch 1 fetched its 7 MB of activations into all eight L2s, 64.8 MB per launch) -- 4 row quarters x 2 column halves
  int mtg, nt;                                // packed m-tile, column tile
  if (!conv_tile_of_block(a.n_tiles_n, a.n_mtiles, a.xcd_rect, blockIdx.x, mtg, nt)) return;   // (padding of an uneven rectangle map)
  const int item = nt / a.tiles_per_item;
  const int tile_in_item = nt - item * a.tiles_per_item;
  constexpr int COLS = SPLITK ? 64 : 256;
  const int t0 = tile_in_item * COLS + (SPLITK ? 0 : wave * 64);
  const int tc = t0 + 2 * li;                 // this lane's column pair (tc, tc+1)
  const bool col_ok = tc < a.T, col_two = tc + 1 < a.T;
  const int row_base = mtg * ROWS;

  int it_begin = 0, it_end = a.n_it;
  if (SPLITK) {
    const int per = (a.n_it + NW - 1) / NW;
    it_begin = wave * per;
    it_end = min(a.n_it, it_begin + per);
  }

  // ---- split-K: the epilogue sites of this wave are static -> their global reads are issued right after the pipeline's
  // first operand loads (so they do not delay the K loop's start) and land behind the K loop.
  // The tile's sites are dealt to the NW waves in order, site s = wave*NS + i:
  //   paired / RB=1: accumulator row r = s & 15;   unpaired RB=2: row block (s >> 4) & 1, r = s & 15.
  constexpr int NS = SPLITK ? (Epi::kPaired ? 16 / NW : RBX * 16 / NW) : 1;
  auto site_row = [&](int sidx) {             // first logical row of the 32-row block the site lives in, + its row inside
    const int blk = sidx >> 4, r = sidx & 15;
    const int rb = Epi::kPaired ? 0 : blk;    // accumulator block = 32-row block of the tile
    return row_base + rb * 32 + acc_row(r, half);
  };
  typename Epi::Pre pre[NS];
  float ln_rsum[PRE == PRE_LNP ? NS : 1];     // PRE_LNP: row sums of the folded weights at this wave's sites (rows padded to the tile: always in range)
  auto prefetch_epilogue = [&]() {
    if constexpr (SPLITK) {
      if (col_ok) {
#pragma unroll
        for (int i = 0; i < NS; ++i) pre[i] = epi.load(item, site_row(wave * NS + i), tc, col_two);
      }
      if constexpr (PRE == PRE_LNP) {
#pragma unroll
        for (int i = 0; i < NS; ++i) ln_rsum[i] = a.ln_R[site_row(wave * NS + i)];
      }
    }
  };

  f32x16 acc[RBX][NB];
#pragma unroll
  for (int x = 0; x < RBX; ++x)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][nb][r] = 0.f;

  // ---- PRE_LN: LayerNorm over the channel (K) axis folded into the GEMM.  The producer (convnext.hip: k_dwconv_stats) stored
  // every channel centred by the mean of ITS group of 32 channels (exact, two-pass) plus the group means and centred sums of
  // squares.  With delta_g = mean_g - mean:  W (u - mean) = W (u - mean_g(c)) + sum_g R[row][g] delta_g,  R = per-group row sums,
  // and var = (sum_g M2_g + 32 sum_g delta_g^2) / K (Chan et al.) -- every term is a product of centred quantities, there is
  // no E[u^2] - mean^2 or acc - mean*rowsum cancellation.  The K loop is untouched; the statistics' loads are issued with the
  // epilogue prefetch and consumed after the reduction.
  constexpr bool kColStats = PRE == PRE_LN || PRE == PRE_RESLN || PRE == PRE_LNP;
  constexpr bool kColMean = PRE == PRE_RESLN || PRE == PRE_LNP;      // the combine stage leaves (mean, rstd) per column instead of (delta_g[16], rstd)
  static_assert(!kColStats || (SPLITK && !Epi::kPaired), "PRE_LN / PRE_RESLN / PRE_LNP: split-K, unpaired epilogues");
  static_assert(PRE != PRE_LN || ROWS * 4 <= NW * 64, "PRE_LN: at most one float4 of the tile's row sums per thread");
  static_assert(PRE != PRE_RESLN || (RB == 1 && std::is_same<Epi, EpiResLN>::value), "PRE_RESLN: 32-row tiles (one statistics group) with EpiResLN");
  __shared__ float ln_rows[PRE == PRE_LN ? ROWS * 16 : 1];   // this tile's rows of ln_R (weights: they come from HBM / MALL)
  __shared__ float ln_cols[kColStats ? 64 * 17 : 1];          // per column of the tile: delta_g[16], rstd
  __shared__ float4 ln_gs[PRE == PRE_RESLN ? NW * 32 : 1];   // PRE_RESLN: per wave and column pair {mean8.x, mean8.y, M2_8.x, M2_8.y} of the new value
  // The statistics are combined ONCE per workgroup, 4 threads per column (thread = column tid >> 2, groups 4q .. 4q+3): two
  // coalesced float4 loads per thread.  (Every lane loading its own two frames' 2 x 128 B -- 16 loads touching 64 different
  // lines each -- kept the CU's L1 tag pipe busy for ~4 us per launch: 29.4 vs 24.7 us for the same GEMM without it.)
  float4 rq{0.f, 0.f, 0.f, 0.f}, sq_mean{0.f, 0.f, 0.f, 0.f}, sq_m2{0.f, 0.f, 0.f, 0.f};
  auto ln_prefetch = [&]() {
    if constexpr (PRE == PRE_LN)   // (32-row tiles: the upper half of the workgroup re-reads the lower half's row sums, unused)
      rq = reinterpret_cast<const float4*>(a.ln_R + (long)row_base * 16)[ROWS * 4 == NW * 64 ? threadIdx.x : threadIdx.x & (ROWS * 4 - 1)];
    if constexpr (kColStats) {
      if (PRE != PRE_RESLN || a.col_stats) {    // (PRE_RESLN without statistics: a plain residual, wave-uniform)
        const int c = threadIdx.x >> 2, q = threadIdx.x & 3;
        const float4* p = reinterpret_cast<const float4*>(a.col_stats + ((long)item * a.T + min(tile_in_item * 64 + c, a.T - 1)) * 32);
        sq_mean = p[q];
        sq_m2 = p[4 + q];
      }
    }
  };
  auto ln_publish = [&]() {   // after the K loop, before the reduction's barrier
    if constexpr (PRE == PRE_LN) {
      if (ROWS * 4 == NW * 64 || threadIdx.x < ROWS * 4) reinterpret_cast<float4*>(ln_rows)[threadIdx.x] = rq;
    }
    if constexpr (kColStats) {
      const int c = threadIdx.x >> 2, q = threadIdx.x & 3;
      const float mg[4] = {sq_mean.x, sq_mean.y, sq_mean.z, sq_mean.w};
      const float m2g[4] = {sq_m2.x, sq_m2.y, sq_m2.z, sq_m2.w};
      float msum = 0.f, m2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * q + k < a.n_groups) { msum += mg[k]; m2 += m2g[k]; }
      msum += __shfl_xor(msum, 1); msum += __shfl_xor(msum, 2);
      m2 += __shfl_xor(m2, 1); m2 += __shfl_xor(m2, 2);
      const float mean = msum / (float)a.n_groups;
      float dev2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = 4 * q + k < a.n_groups ? mg[k] - mean : 0.f;
        if constexpr (!kColMean) ln_cols[c * 17 + 4 * q + k] = d;
        dev2 += d * d;
      }
      dev2 += __shfl_xor(dev2, 1); dev2 += __shfl_xor(dev2, 2);
      if (q == 0) {
        ln_cols[c * 17 + 16] = 1.f / sqrtf((m2 + 32.f * dev2) / (float)(32 * a.n_groups) + a.ln_eps);
        if constexpr (kColMean) ln_cols[c * 17] = mean;
      }
    }
  };

  const bool active = SPLITK ? true : (t0 < a.T);   // whole-wave overhang tiles skip the K loop
  if constexpr (OPK == OPK_BF16) {
    // ---- bf16 operands: one K iteration = 16 channels of one tap = ONE v_mfma_f32_32x32x16_bf16 per accumulator block.
    // A: packed [m_tile][it][rb][lane] 16 B = 8 bf16 (k = 8*half .. +7 of row li); B: the lane's column, channel block
    // 2*cb16 + half, one 16-byte group.  Same register ring and saturating cursors as the fp32 loop below.
    static_assert(PRE == PRE_NONE, "bf16 operands: plain contraction only");
    if (active && it_begin < it_end) {
      struct StageB { bf16x8 a[RBX]; bf16x8 b[NB]; };
      const int n = it_end - it_begin;
      const int cb0 = it_begin / a.taps, tap0 = it_begin - cb0 * a.taps;
      const char* Abase = reinterpret_cast<const char*>(a.Wp + ((long)mtg * a.n_it + it_begin) * (RB * 64));
      const char* Xbase = reinterpret_cast<const char*>(a.X + item * a.x_bstride) + (long)(a.shift0 + t0) * 16;
      const unsigned rs = (unsigned)a.ldx * 16u;                 // bytes between 8-channel blocks
      const unsigned d_tap = (unsigned)a.dshift * 16u;
      const unsigned d_wrap = 2u * rs - (unsigned)(a.taps - 1) * d_tap;
      const int itl = it_end - 1, cbl = itl / a.taps, tapl = itl - cbl * a.taps;
      const unsigned a_last = (unsigned)(n - 1) * (RB * 1024u);
      const unsigned x_last = (unsigned)cbl * 2u * rs + (unsigned)tapl * d_tap;
      unsigned a_off = 0, x_off = (unsigned)cb0 * 2u * rs + (unsigned)tap0 * d_tap;
      int tap = tap0;
      const unsigned a_lane = lane * 16u;
      unsigned x_lane[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) x_lane[nb] = (unsigned)half * rs + (unsigned)(2 * li + nb) * 16u;
      auto load = [&](StageB& s) {
#pragma unroll
        for (int x = 0; x < RBX; ++x) s.a[x] = *reinterpret_cast<const bf16x8*>(Abase + (a_off + a_lane + x * 1024u));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) s.b[nb] = *reinterpret_cast<const bf16x8*>(Xbase + (x_off + x_lane[nb]));
        const bool wrap = tap + 1 == a.taps;
        a_off = min(a_off + RB * 1024u, a_last);
        x_off = min(x_off + (wrap ? d_wrap : d_tap), x_last);
        tap = wrap ? 0 : tap + 1;
      };
      auto compute = [&](StageB& s) {
#pragma unroll
        for (int x = 0; x < RBX; ++x)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.a[x], s.b[nb], acc[x][nb], 0, 0, 0);
      };
      constexpr int D = 4;
      StageB st[D];
#pragma unroll
      for (int d = 0; d < D - 1; ++d) load(st[d]);
      __builtin_amdgcn_sched_barrier(0);
      prefetch_epilogue();
      __builtin_amdgcn_sched_barrier(0);
      int done = 0;
      for (; done + D <= n; done += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          load(st[(d + D - 1) % D]);
          compute(st[d]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int d = 0; d < D - 1; ++d)
        if (done + d < n) compute(st[d]);
    } else {
      prefetch_epilogue();
    }
  } else
  if (active && it_begin < it_end) {
    struct Stage { float4 a[RBX]; f2 b[4]; };
    // Wave-uniform bases (SGPR pairs) + 32-bit byte cursors (SGPR) + per-lane 32-bit byte offsets (VGPR).  The cursors
    // saturate at the wave's last K iteration, so the pipeline's run-ahead loads are unconditional and never leave
    // this wave's K range (the over-run re-reads the last iteration: L1 hits, values unused).
    const int n = it_end - it_begin;
    const int cb0 = it_begin / a.taps, tap0 = it_begin - cb0 * a.taps;
    const char* Abase = reinterpret_cast<const char*>(a.Wp + ((long)mtg * a.n_it + it_begin) * (RB * 64));
    const char* Xbase = reinterpret_cast<const char*>(a.X + item * a.x_bstride + a.shift0 + t0);
    const unsigned rs = (unsigned)a.ldx * 4u;                  // bytes between channels
    const unsigned d_tap = (unsigned)a.dshift * 4u;            // next tap, same channel block
    const unsigned d_wrap = 8u * rs - (unsigned)(a.taps - 1) * d_tap;   // first tap of the next channel block
    const int itl = it_end - 1, cbl = itl / a.taps, tapl = itl - cbl * a.taps;
    const unsigned a_last = (unsigned)(n - 1) * (RB * 1024u);
    const unsigned x_last = (unsigned)cbl * 8u * rs + (unsigned)tapl * d_tap;
    unsigned a_off = 0, x_off = (unsigned)cb0 * 8u * rs + (unsigned)tap0 * d_tap;
    int tap = tap0;
    const unsigned a_lane = lane * 16u;
    unsigned x_lane[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x_lane[j] = (unsigned)(half * 4 + j) * rs + (unsigned)li * 8u;

    auto load = [&](Stage& s) {
#pragma unroll
      for (int x = 0; x < RBX; ++x)
        s.a[x] = *reinterpret_cast<const float4*>(Abase + (a_off + a_lane + x * 1024u));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f2u v = *reinterpret_cast<const f2u*>(Xbase + (x_off + x_lane[j]));
        s.b[j] = f2{v.x, v.y};
      }
      const bool wrap = tap + 1 == a.taps;
      a_off = min(a_off + RB * 1024u, a_last);
      x_off = min(x_off + (wrap ? d_wrap : d_tap), x_last);
      tap = wrap ? 0 : tap + 1;
    };
    auto compute = [&](Stage& s) {
      if (PRE == PRE_LRELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s.b[j].x = s.b[j].x > 0.f ? s.b[j].x : s.b[j].x * a.in_slope;
          s.b[j].y = s.b[j].y > 0.f ? s.b[j].y : s.b[j].y * a.in_slope;
        }
      }

#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int x = 0; x < RBX; ++x) {
          const float av = j == 0 ? s.a[x].x : j == 1 ? s.a[x].y : j == 2 ? s.a[x].z : s.a[x].w;
          acc[x][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s.b[j].x, acc[x][0], 0, 0, 0);
          acc[x][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s.b[j].y, acc[x][1], 0, 0, 0);
        }
      }
    };
    // One pipeline slot = "issue the loads of stage L, run the MFMAs of stage C".  The loads and the cursor arithmetic
    // are spread between the first MFMAs (a wave issues in order: anything placed in one lump between two MFMA groups
    // leaves the matrix pipe idle for as long as it takes to issue).
    auto slot = [&](Stage& L, Stage& C) {
      load(L);
      compute(C);
#pragma unroll
      for (int k = 0; k < RBX + 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // a few VALU / SALU
      }
      __builtin_amdgcn_sched_group_barrier(0x008, RBX * 8 - (RBX + 4), 0);
      __builtin_amdgcn_sched_barrier(0);
    };

    // Software pipeline over a D-stage register ring: the operands of iteration i+D-1 are requested while iteration i
    // runs.  Both operand streams arrive from beyond the XCD's L2 on first touch (weights from HBM / Infinity Cache,
    // activations from the L2 of whichever XCD wrote them), and all workgroups sharing a line run in lockstep, so the
    // latency to cover is the ~2-4 k cycle fabric latency, not an L2 hit: D-1 = 3 slots of 1024 MFMA cycles (D = 3, 4
    // and 6 measure within 2 % of each other; the residual ~12 % gap to the no-load MFMA rate is not latency).
    // No conditional loads and no register copies: the only waits are counted vmcnt(D-1 stages in flight).
    constexpr int D = 4;
    Stage st[D];
    FDX_STAMP(1);
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load(st[d]);
    __builtin_amdgcn_sched_barrier(0);
    int done = 0;
    prefetch_epilogue();
    ln_prefetch();
    __builtin_amdgcn_sched_barrier(0);
    for (; done + D <= n; done += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) slot(st[(d + D - 1) % D], st[d]);
#ifdef FDX_KTRACE
      if (done == 0) FDX_STAMP(6);   // first pass over the ring done: (t6 - t1) - D slots of MFMA issue = the pipeline's fill time
#endif
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
      if (done + d < n) compute(st[d]);
  } else {
    prefetch_epilogue();
    ln_prefetch();
  }
  FDX_STAMP(2);

  if (SPLITK) {
    // ---- cross-wave K reduction through LDS, fixed summation order w0 + w1 + ... (deterministic).
    // Layout red[wave][r][lane][x*NB + nb]: a lane's V values of accumulator row r are contiguous -> 16-byte LDS
    // accesses, conflict-free (consecutive lanes, consecutive words).
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto ridx = [&](int w, int r) { return ((w * 16 + r) * kWave + lane) * V; };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (V == 2) {
        *reinterpret_cast<f2*>(red + ridx(wave, r)) = f2{acc[0][0][r], acc[0][1][r]};
      } else {
#pragma unroll
        for (int q = 0; q < V / 4; ++q)   // block pair (x = 2q, 2q+1): {x nb0, x nb1, x+1 nb0, x+1 nb1}
          *reinterpret_cast<f4*>(red + ridx(wave, r) + 4 * q) =
              f4{acc[2 * q][0][r], acc[2 * q][1][r], acc[2 * q + 1][0][r], acc[2 * q + 1][1][r]};
      }
    }
    ln_publish();
    FDX_STAMP(3);
    __syncthreads();
    FDX_STAMP(4);
    if constexpr (PRE == PRE_RESLN) {
      // ---- EpiResLN: new residual value and its 32-row group statistics across the four waves.  A wave holds 8 of the
      // group's rows per column (4 per lane half): two-pass (mean8, M2_8) inside the wave, ONE exchange, Chan's combination in wave order.
      static_assert(NS == 4, "a wave's four sites = four consecutive rows per lane half");
      const float* lc = ln_cols + (2 * li) * 17;
      const int g = mtg;                                   // ROWS == 32: the tile's rows are statistics group mtg
      const f2 dl{-lc[0], -lc[17]}, rstd{lc[16], lc[17 + 16]};   // (old - mean) rstd
      f2 v[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int r = (wave * NS + i) & 15;
        f2 sum = *reinterpret_cast<const f2*>(red + ridx(0, r));
#pragma unroll
        for (int w = 1; w < NW; ++w) sum += *reinterpret_cast<const f2*>(red + ridx(w, r));
        v[i] = epi.value(sum, pre[i], dl, rstd);
      }
      f2 s8 = (v[0] + v[1]) + (v[2] + v[3]);
      s8.x += __shfl_xor(s8.x, 32); s8.y += __shfl_xor(s8.y, 32);
      const f2 mean8 = s8 * 0.125f;
      f2 m8{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NS; ++i) { const f2 d = v[i] - mean8; m8 += d * d; }
      m8.x += __shfl_xor(m8.x, 32); m8.y += __shfl_xor(m8.y, 32);
      if (half == 0) ln_gs[wave * 32 + li] = float4{mean8.x, mean8.y, m8.x, m8.y};
      __syncthreads();
      const float4 p0 = ln_gs[li], p1 = ln_gs[32 + li], p2 = ln_gs[64 + li], p3 = ln_gs[96 + li];
      const f2 mean_g = f2{((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y} * 0.25f;
      const f2 e0 = f2{p0.x, p0.y} - mean_g, e1 = f2{p1.x, p1.y} - mean_g, e2 = f2{p2.x, p2.y} - mean_g, e3 = f2{p3.x, p3.y} - mean_g;
      const f2 M2 = ((f2{p0.z, p0.w} + f2{p1.z, p1.w}) + f2{p2.z, p2.w}) + f2{p3.z, p3.w} + 8.f * (((e0 * e0 + e1 * e1) + e2 * e2) + e3 * e3);
      if (!col_ok) return;
#pragma unroll
      for (int i = 0; i < NS; ++i)
        st2p_keep(epi.X + item * epi.bs + (long)site_row(wave * NS + i) * epi.ld + tc, v[i], col_two);
      if (wave == 0 && half == 0) {
        float* st = epi.st_out + ((long)item * epi.T + tc) * 32 + g;
        st[0] = mean_g.x; st[16] = M2.x;
        if (col_two) { st[32] = mean_g.y; st[48] = M2.y; }
      }
      FDX_STAMP(5);
      return;
    }
    if (!col_ok) return;
    if constexpr (Epi::kPaired) {
      f4 sum[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int r = (wave * NS + i) & 15;
        sum[i] = *reinterpret_cast<const f4*>(red + ridx(0, r));
#pragma unroll
        for (int w = 1; w < NW; ++w) sum[i] += *reinterpret_cast<const f4*>(red + ridx(w, r));
      }
      if constexpr (OPK == OPK_BF16) {
        static_assert(NS == 4, "a wave's four sites = one channel quad");
        f2 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = epi.value(f2{sum[i][0], sum[i][1]}, f2{sum[i][2], sum[i][3]}, pre[i]);
        epi.store4(item, site_row(wave * NS), tc, col_two, v);
      } else {
#pragma unroll
        for (int i = 0; i < NS; ++i)
          epi.store(item, site_row(wave * NS + i), tc, col_two, f2{sum[i][0], sum[i][1]}, f2{sum[i][2], sum[i][3]}, pre[i]);
      }
    } else {
      f2 sum[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int sidx = wave * NS + i, blk = sidx >> 4, r = sidx & 15;   // blk = accumulator block x
        sum[i] = *reinterpret_cast<const f2*>(red + ridx(0, r) + 2 * blk);
#pragma unroll
        for (int w = 1; w < NW; ++w) sum[i] += *reinterpret_cast<const f2*>(red + ridx(w, r) + 2 * blk);
      }
      if constexpr (PRE == PRE_LNP) {
        const float* lc = ln_cols + (2 * li) * 17;
        const f2 mean{lc[0], lc[17]}, rstd{lc[16], lc[17 + 16]};
#pragma unroll
        for (int i = 0; i < NS; ++i) sum[i] = (sum[i] - mean * ln_rsum[i]) * rstd;
      }
      if constexpr (PRE == PRE_LN) {
        f2 delta[16];
        const float* lc = ln_cols + (2 * li) * 17;   // this lane's column pair inside the tile
#pragma unroll
        for (int g = 0; g < 16; ++g) delta[g] = f2{lc[g], lc[17 + g]};
        const f2 rstd{lc[16], lc[17 + 16]};
        constexpr int HS = NS >= 4 ? NS / 2 : NS;   // two batches of row-sum reads (LDS): all NS rows at once would not fit 256 VGPRs
#pragma unroll
        for (int h0 = 0; h0 < NS; h0 += HS) {
          float4 R[HS][4];
#pragma unroll
          for (int i = 0; i < HS; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) R[i][q] = reinterpret_cast<const float4*>(ln_rows + (site_row(wave * NS + h0 + i) - row_base) * 16)[q];
#pragma unroll
          for (int i = 0; i < HS; ++i) {
            f2 corr{0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              corr += R[i][q].x * delta[4 * q] + R[i][q].y * delta[4 * q + 1];
              corr += R[i][q].z * delta[4 * q + 2] + R[i][q].w * delta[4 * q + 3];
            }
            sum[h0 + i] = (sum[h0 + i] + corr) * rstd;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (OPK == OPK_BF16) {
        static_assert(NS == 8, "a wave's eight sites = two channel quads");
        f2 y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { y[i] = f2{0.f, 0.f}; epi.store(item, site_row(wave * NS + i), tc, col_two, sum[i], pre[i], y[i]); }
        const f2 (&ya)[4] = *reinterpret_cast<const f2 (*)[4]>(&y[0]);
        const f2 (&yb)[4] = *reinterpret_cast<const f2 (*)[4]>(&y[4]);
        epi.store_y4(item, site_row(wave * NS), tc, col_two, ya);
        epi.store_y4(item, site_row(wave * NS + 4), tc, col_two, yb);
      } else {
#pragma unroll
        for (int i = 0; i < NS; ++i) epi.store(item, site_row(wave * NS + i), tc, col_two, sum[i], pre[i]);
      }
    }
    FDX_STAMP(5);
  } else {
    if (!active || !col_ok) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (Epi::kPaired) {
        const int row = row_base + acc_row(r, half);
        epi.store(item, row, tc, col_two, f2{acc[0][0][r], acc[0][1][r]}, f2{acc[1][0][r], acc[1][1][r]},
                  epi.load(item, row, tc, col_two));
      } else {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const int row = row_base + rb * 32 + acc_row(r, half);
          epi.store(item, row, tc, col_two, f2{acc[rb][0][r], acc[rb][1][r]}, epi_load_late(epi, item, row, tc, col_two));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ host launch
// Tile -> XCD map of the split-K family (small grids).  Row runs fetch the weights once chip-wide and the activation operand once per XCD: right
// while there are more rows than columns.  With more columns than rows (ConvNext pwconv2, the transformer's linear2 / out-projections at batch 1:
// 512 rows x 861 columns) 4 row quarters x 2 column halves halve the activation re-reads for 4 x the (smaller) weight reads.  Needs the row tiles
// to divide by 4 (an odd column-tile count pads the launch, round 6); FDX_SPLITK_RECT=0 keeps row runs everywhere (A/B).  A scheduling choice only:
// results are bit-identical.
inline int splitk_xcd_rect(int n_tiles_n, int n_mt, long rows, long cols) {
  static const int on = [] { const char* e = getenv("FDX_SPLITK_RECT"); return e ? atoi(e) : 1; }();
  if (!on || cols <= rows) return 0;
  return ((n_mt & 3) == 0 && n_tiles_n >= 2 && (long)n_tiles_n * n_mt >= 64) ? 2 : 0;   // (odd column-tile counts: padded launch, conv_rect_grid)
}

struct ConvGeom {   // everything the launcher needs besides pointers
  int B, T;         // items, valid columns per item
  int cin8;         // Cin / 8 (padded)
  int taps, shift0, dshift;
  int n_mtiles;     // row tiles of 32*RB logical rows (32 pairs for paired epilogues)
};

template <int RB, bool SPLITK, int PRE, class Epi, int OPK = OPK_F32>
inline hipError_t launch_convgemm(const ConvGeom& g, const float4* Wp, const float* X, long x_bstride, int ldx,
                                  float in_slope, const Epi& epi, hipStream_t s, hipEvent_t ev_start = nullptr,
                                  hipEvent_t ev_stop = nullptr, const float* col_stats = nullptr, const float* ln_R = nullptr,
                                  int n_groups = 0, float ln_eps = 0.f) {
  ConvArgs a;
  a.Wp = Wp; a.X = X; a.x_bstride = x_bstride; a.ldx = ldx;
  a.n_it = g.cin8 * g.taps; a.taps = g.taps; a.shift0 = g.shift0; a.dshift = g.dshift;
  a.T = g.T;
  const int cols = SPLITK ? 64 : 256;
  a.tiles_per_item = (g.T + cols - 1) / cols;
  a.n_tiles_n = g.B * a.tiles_per_item;
  a.n_mtiles = g.n_mtiles;
  a.xcd_rect = SPLITK ? splitk_xcd_rect(a.n_tiles_n, a.n_mtiles, 32 * RB * a.n_mtiles, (long)g.B * g.T) : 0;
  a.in_slope = in_slope;
  a.col_stats = col_stats; a.ln_R = ln_R; a.n_groups = n_groups; a.ln_eps = ln_eps;
  const int grid = conv_rect_grid(a.n_tiles_n, a.n_mtiles, a.xcd_rect);
  if (grid <= 0) return hipSuccess;
#ifdef FDX_KTRACE
  a.trace = nullptr;
  if (g_trace.buf && g_trace.n < g_trace.max_launches && grid <= g_trace.blocks_cap)
    a.trace = g_trace.buf + (size_t)(g_trace.n++) * g_trace.blocks_cap * 32;
#endif
  if (ev_start)   // profiling: the events receive this dispatch's own begin / end timestamps (what rocprofv3 reports)
    hipExtLaunchKernelGGL((convgemm_kernel<RB, SPLITK, PRE, Epi, OPK>), dim3(grid), dim3(256), 0, s, ev_start, ev_stop, 0,
                          FDX_CONV_HOT_ARGS(a), conv_cold_of(a), epi);
  else
    hipLaunchKernelGGL((convgemm_kernel<RB, SPLITK, PRE, Epi, OPK>), dim3(grid), dim3(256), 0, s, FDX_CONV_HOT_ARGS(a), conv_cold_of(a), epi);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ host packing
// dst[((mt*n_it + cb*taps + tap)*RB + rb)*64 + lane][j] = w(row(mt, rb, lane&31), c = cb*8 + (lane>>5)*4 + j, tap)
// `getw(mt, rb, i, c, tap)` returns the logical weight (0 for padding).
template <class F>
inline void pack_convgemm(float* dst, int n_mtiles, int RB, int cin8, int taps, F getw) {
  const int n_it = cin8 * taps;
  for (int mt = 0; mt < n_mtiles; ++mt)
    for (int cb = 0; cb < cin8; ++cb)
      for (int tap = 0; tap < taps; ++tap)
        for (int rb = 0; rb < RB; ++rb) {
          float* d = dst + ((((size_t)mt * n_it + (size_t)cb * taps + tap) * RB + rb) * 64) * 4;
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j)
              d[lane * 4 + j] = getw(mt, rb, lane & 31, cb * 8 + (lane >> 5) * 4 + j, tap);
        }
}
inline size_t packed_floats(int n_mtiles, int RB, int cin8, int taps) {
  return (size_t)n_mtiles * cin8 * taps * RB * 64 * 4;
}

}  // namespace fdx
