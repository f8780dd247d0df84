// convgemm16s.hip.h -- the residual-block pair of the denoiser on `v_mfma_f32_16x16x4_f32` with a SHAPE-ADAPTIVE workgroup tile.
//
// The denoiser's two GEMMs are small at batch 1 (1024 rows x ~861 columns): with a fixed 64 x 64 tile they give 224 workgroups
// for 256 CUs, and at 5 s (T = 430) only 112.  Per-element arithmetic on this MFMA does not depend on the tile a workgroup
// owns (every output is the same k-ordered fp32 fma chain, split over the same four K ranges and summed in the same order), so
// the tile shape is a pure scheduling choice -- made per launch by `pick_shape16` from (rows, batch, frames):
//
//   tile = NR 16-row blocks  x  NM*16 columns          NR in {2, 4},  NM in {4 .. 8}
//   lane l: lj = l & 15 owns the NM ADJACENT columns t0 + NM*lj .. +NM-1 (one 16-byte load + one tail load per k-row feeds NM
//   MFMAs of interleaved column sets {NM*j + m}), lk = l >> 4 is the k-row of B / k of A / row quad of D.
//   T = 861:  NR = 2, NM = 7  ->  32 x 8 = 256 workgroups of 32 x 112 (was 16 x 14 = 224 of 64 x 64): 7/8 of the MFMA work
//   per workgroup and every CU busy.   T = 430: NR = 2, NM = 4 -> 224 workgroups (was 112).
//
// Same contraction, split-K workgroup (4 waves, fixed-order LDS reduction), operand pipeline and epilogue arithmetic as
// convgemm16.hip.h (whose NR = 4, NM = 4 instance this file generalises).
//
// A fragments are host-packed for NR = 4 (`pack_convgemm16`: one float4 per lane = the four 16-row blocks of a 64-row tile);
// the NR = 2 order (one float2 per lane) is derived from it on the device at attach time (`k_repack16_nr2`), so the packed
// arena -- what the RCCL broadcast ships -- does not change.
#pragma once
#include "convgemm16.hip.h"

namespace fdx {

template <int N> struct VecN { float v[N]; };
struct __attribute__((packed, aligned(4))) f3u { float x, y, z; };

// The same N adjacent floats kept as the VECTORS the loads produce (one dwordx4 + one tail load): the K loop's operand ring holds
// these, so that its loop-carried values are register tuples, not 7-8 scalars hipcc then has to re-assemble (at N >= 7 it copied all
// three pre-loaded stages into other registers in front of the loop, behind a full s_waitcnt).
typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));
typedef float f3a __attribute__((ext_vector_type(3), aligned(4)));
typedef float f2a __attribute__((ext_vector_type(2), aligned(4)));
template <int N> struct RawTail { typedef float type; };
template <> struct RawTail<6> { typedef f2u type; };   // (as a 2-vector hipcc routed this instance's ring through scratch memory)
template <> struct RawTail<7> { typedef f3a type; };
template <> struct RawTail<8> { typedef f4a type; };
template <int N> struct RawN {
  f4a lo; typename RawTail<N>::type hi;
  template <int M> __device__ __forceinline__ float get() const {
    if constexpr (M < 4) return lo[M];
    else if constexpr (N == 5) return hi;
    else if constexpr (N == 6) return M == 4 ? hi.x : hi.y;
    else return hi[M - 4];
  }
};
template <int N> __device__ __forceinline__ RawN<N> ldRaw(const float* p) {
  RawN<N> r;
  r.lo = *reinterpret_cast<const f4a*>(p);
  if constexpr (N == 4) r.hi = 0.f;
  else r.hi = *reinterpret_cast<const typename RawTail<N>::type*>(p + 4);
  return r;
}

// N adjacent floats at any dword alignment (sources are the library's own padded rows: >= kTailPad floats of slack right of T)
template <int N> __device__ __forceinline__ VecN<N> ldN(const float* p) {
  static_assert(N >= 4 && N <= 8, "4..8 columns per lane");
  VecN<N> r;
  const f4u a = *reinterpret_cast<const f4u*>(p);
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  if constexpr (N == 5) { r.v[4] = p[4]; }
  if constexpr (N == 6) { const f2u b = *reinterpret_cast<const f2u*>(p + 4); r.v[4] = b.x; r.v[5] = b.y; }
  if constexpr (N == 7) { const f3u b = *reinterpret_cast<const f3u*>(p + 4); r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; }
  if constexpr (N == 8) { const f4u b = *reinterpret_cast<const f4u*>(p + 4); r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w; }
  return r;
}
// store into a padded row: columns >= T (at most N-1, inside the right pad) receive 0 -- what the pad must hold anyway.
// NT: non-temporal (tiles read next by other XCDs), else normal caching (tiles the same workgroup re-reads next layer).
template <int N, bool NT> __device__ __forceinline__ void stNp(float* p, VecN<N> v, int nvalid) {
#pragma unroll
  for (int i = 1; i < N; ++i) v.v[i] = nvalid > i ? v.v[i] : 0.f;
  typedef float f4nt __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f3nt __attribute__((ext_vector_type(3), aligned(4)));
  typedef float f2nt __attribute__((ext_vector_type(2), aligned(4)));
#if defined(FDX_BISECT) && FDX_BISECT == 2
  constexpr bool kNT = false;                                          // (clock bisect: plain stores everywhere)
#else
  constexpr bool kNT = NT;
#endif
  if constexpr (kNT) {
    __builtin_nontemporal_store(f4nt{v.v[0], v.v[1], v.v[2], v.v[3]}, reinterpret_cast<f4nt*>(p));
    if constexpr (N == 5) __builtin_nontemporal_store(v.v[4], p + 4);
    if constexpr (N == 6) __builtin_nontemporal_store(f2nt{v.v[4], v.v[5]}, reinterpret_cast<f2nt*>(p + 4));
    if constexpr (N == 7) __builtin_nontemporal_store(f3nt{v.v[4], v.v[5], v.v[6]}, reinterpret_cast<f3nt*>(p + 4));
    if constexpr (N == 8) __builtin_nontemporal_store(f4nt{v.v[4], v.v[5], v.v[6], v.v[7]}, reinterpret_cast<f4nt*>(p + 4));
  } else {
    f4u a; a.x = v.v[0]; a.y = v.v[1]; a.z = v.v[2]; a.w = v.v[3];
    *reinterpret_cast<f4u*>(p) = a;
    if constexpr (N == 5) p[4] = v.v[4];
    if constexpr (N == 6) { f2u b; b.x = v.v[4]; b.y = v.v[5]; *reinterpret_cast<f2u*>(p + 4) = b; }
    if constexpr (N == 7) { f3u b; b.x = v.v[4]; b.y = v.v[5]; b.z = v.v[6]; *reinterpret_cast<f3u*>(p + 4) = b; }
    if constexpr (N == 8) { f4u b; b.x = v.v[4]; b.y = v.v[5]; b.z = v.v[6]; b.w = v.v[7]; *reinterpret_cast<f4u*>(p + 4) = b; }
  }
}

// Right pad (floats) every row of the denoiser's activation buffers needs for these tiles: the last tile may overhang T by up to
// 16*8 - 1 columns and a dilated tap reads up to 16 columns further.
constexpr int kTailPad = 160;

template <int NM> struct EpiGate16S {  // wavenet.py:112-115 (EpiGate16 for NM columns per lane)
  static constexpr bool kPaired = true;
  float* out; long o_bs; int ldo;
  const float* P; long p_bs; int ldp;
  int C;
  struct Pre { VecN<NM> pg, pf; };
  __device__ __forceinline__ Pre load(int b, int row, int t) const {
    const float* q = P + b * p_bs + t;
    return Pre{ldN<NM>(q + (long)row * ldp), ldN<NM>(q + (long)(row + C) * ldp)};
  }
  __device__ __forceinline__ void store(int b, int row, int t, int nvalid, const VecN<NM>& g, const VecN<NM>& f, const Pre& p) const {
    VecN<NM> z;
#pragma unroll
#if defined(FDX_BISECT) && FDX_BISECT == 1
    for (int m = 0; m < NM; ++m) z.v[m] = (g.v[m] + p.pg.v[m]) + (f.v[m] + p.pf.v[m]);     // (clock bisect: no exp / rcp)
#else
    for (int m = 0; m < NM; ++m) z.v[m] = EpiGate::gate1(g.v[m] + p.pg.v[m], f.v[m] + p.pf.v[m]);
#endif
    stNp<NM, true>(out + b * o_bs + (long)row * ldo + t, z, nvalid);
  }
};

template <int NM> struct EpiResSkip16S {  // wavenet.py:117-120 + the skip sum of :228 (EpiResSkip16 for NM columns per lane)
  static constexpr bool kPaired = false;
  float* X; float* Y; float* SK; long bs; int ld;
  const float* bias;
  const float* sb; int sb_ld, sb_bs;
  int C, skip_mode;
  float inv_div, r_inv_div;
  // exact-mask mode (fdx_sampler_run_ragged): keep[b][t] = 1 where a frame exists, 0 where it does not (padded row layout, 0 in the
  // pads); Y -- the next dilated conv's input -- is written as 0 wherever keep is 0, so that frames next to a hole read zeros exactly
  // like the conv's own zero padding.  null: every column of [0, T) exists.
  const float* keep; long keep_bs;
  struct Pre { VecN<NM> old; VecN<NM> keep; float bias, sb; };
  __device__ __forceinline__ bool is_res(int row) const { return __builtin_amdgcn_readfirstlane(row) < C; }
  // Branch-free: every load is unconditional, from a pointer selected by the (wave-uniform) conditions.  With the loads inside
  // `if (res) .. else if (skip) ..` hipcc put an `s_waitcnt vmcnt(0)` in front of each epilogue site's loads (the two branches
  // write the same registers), i.e. four full memory round trips in a row -- with the K loop's first operand stages in flight
  // behind them -- before the first MFMA: most of the 4.2 k cycles "set-up + prefetch" of this kernel at batch 1.
  __device__ __forceinline__ Pre load(int b, int row, int t) const {
    Pre p;
    const bool res = is_res(row), use_y = res && Y, use_keep = use_y && keep;
    const float* src = res ? X + b * bs + (long)row * ld + t : SK + b * bs + (long)(row - C) * ld + t;   // (SK unread by store() when the sum starts here)
    const float* sbp = use_y ? sb + (long)row * sb_ld + b * sb_bs : bias + row;
    const float* kp = use_keep ? keep + b * keep_bs + t : bias;      // (any 8 readable floats; one line for all lanes)
    p.old = ldN<NM>(src);
    p.bias = bias[row];
    p.sb = *sbp;
    p.keep = ldN<NM>(kp);                       // (raw: store() applies `use_keep` -- touching a loaded value here would wait for it)
    return p;
  }
  __device__ __forceinline__ void store(int b, int row, int t, int nvalid, VecN<NM> v, const Pre& p) const {
#pragma unroll
    for (int m = 0; m < NM; ++m) v.v[m] += p.bias;
    if (is_res(row)) {
      const long o = b * bs + (long)row * ld + t;
      VecN<NM> xn, yn;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        xn.v[m] = div_const(p.old.v[m] + v.v[m], 1.41421356237309504880f, 0.70710678118654752440f);
        yn.v[m] = (!keep || p.keep.v[m] != 0.f) ? xn.v[m] + p.sb : 0.f;
      }
      stNp<NM, false>(X + o, xn, nvalid);
      if (Y) stNp<NM, true>(Y + o, yn, nvalid);
    } else {
      const long o = b * bs + (long)(row - C) * ld + t;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        float s = v.v[m];
        if (skip_mode == 1 || skip_mode == 2) s = p.old.v[m] + s;
        if (skip_mode >= 2) s = div_const(s, inv_div, r_inv_div);
        v.v[m] = s;
      }
      stNp<NM, false>(SK + o, v, nvalid);
    }
  }
};

// out = act(acc + bias[row]) for NM adjacent columns (EpiBias of convgemm.hip.h, the part the K = 512 GEMMs of the widening denoisers use: no
// mask, no second output).  The result is the next GEMM's B operand, read by every XCD: non-temporal stores.
template <int NM> struct EpiBiasAct16S {
  static constexpr bool kPaired = false;
  float* out; long o_bs; int ldo;
  const float* bias;
  int act;
  struct Pre { float bias; };
  __device__ __forceinline__ Pre load(int /*b*/, int row, int /*t*/) const { return Pre{bias[row]}; }
  __device__ __forceinline__ void store(int b, int row, int t, int nvalid, VecN<NM> v, const Pre& p) const {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      float y = v.v[m] + p.bias;
      if (act == ACT_RELU) y = fmaxf(y, 0.f);
      else if (act == ACT_MISH) y = mish_f(y);
      else if (act == ACT_GELU) y = gelu_f(y);
      v.v[m] = y;
    }
    stNp<NM, true>(out + b * o_bs + (long)row * ldo + t, v, nvalid);
  }
};

// ------------------------------------------------------------------------------------------ kernel
// Packed A: NR = 4: [m64-tile][it][h][lane] float4 (pack_convgemm16);  NR = 2: [m32-tile][it][h][lane] float2 (k_repack16_nr2);
// NR = 1 (out-projection only): [m16-tile][it][h][lane] float (k_repack16_from32<1>) -- 16 x (16 NM) tiles, twice the workgroups of NR = 2.
// Paired epilogues: blocks 0 .. NR/2-1 hold the tile's gate rows (16 channels each), blocks NR/2 .. NR-1 the matching filter rows.
// (An LDS-staged operand path for this K loop -- per-wave LDS-DMA rings, asm-sequenced MFMA / ds_read stream -- was built and measured in
// round 3: bit-identical, fewer cycles per wave, LONGER launches (26.6 vs 25.2 us); removed in round 4, see profiles/NOTES.md and commit 7078c9b.)
// PRE = PRE_LNP (round 6, second session: the K = 512 GEMMs of the ConvNext block on this family): LayerNorm over K folded in with the B operand
// left as it is -- result = rstd[t] (acc - mean[t] rowsum[row]) in front of the epilogue, exactly convgemm.hip.h's PRE_LNP: a.col_stats =
// [item][T][2][16] group means / M2s (combined per column ONCE per workgroup, same arithmetic), a.ln_R = [rows] row sums of the folded weights.
template <class Epi, int NR, int NM, int PRE = PRE_NONE>
__global__ __launch_bounds__(256) void convgemm16s_kernel(FDX_CONV_HOT_PARAMS, ConvArgsCold cold, Epi epi) {
  FDX_CONV_ARGS_FROM_HOT(cold);
  static_assert(NR == 2 || NR == 4 || (NR == 1 && !Epi::kPaired), "1 (unpaired epilogues only), 2 or 4 row blocks");
  static_assert(PRE == PRE_NONE || (PRE == PRE_LNP && !Epi::kPaired), "plain, or PRE_LNP with an unpaired epilogue");
  constexpr int NW = 4, COLS = 16 * NM, STR = NM <= 4 ? 4 : 8;      // LDS stride (floats) of a lane's NM partial sums
  a.tiles_per_item = (a.T + COLS - 1) / COLS;
  __shared__ float red[NW * NR * 4 * kWave * STR];                  // [wave][x*4 + reg][lane][m]
  constexpr bool kLnp = PRE == PRE_LNP;
  constexpr int LNPASS = (COLS + 63) / 64;                          // 4 threads per column, 64 columns per pass
  __shared__ float ln_cols[kLnp ? COLS * 2 : 1];                    // per column of the tile: mean, rstd

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lj = lane & 15, lk = lane >> 4;
  FDX_STAMP(0);
  FDX_STAMP_RT0();

  int mt, nt;
  if (!conv_tile_of_block(a.n_tiles_n, a.n_mtiles, a.xcd_rect, blockIdx.x, mt, nt)) return;   // (padding of an uneven rectangle map)
  const int item = nt / a.tiles_per_item;
  const int t0 = (nt - item * a.tiles_per_item) * COLS;
  const int tc = t0 + NM * lj;                                        // this lane's first column
  const int nvalid = min(NM, a.T - tc);                               // <= 0: all of the lane's columns are overhang
  constexpr int ROWS = Epi::kPaired ? NR * 8 : NR * 16;               // logical rows (pairs) per tile
  const int row_base = mt * ROWS;

  const int per = (a.n_it + NW - 1) / NW;
  const int it_begin = wave * per, it_end = min(a.n_it, it_begin + per);

  // epilogue sites of the tile: paired: (gate block gb, reg) -> NR/2 * 4;  unpaired: (block x, reg) -> NR * 4; NS per wave
  constexpr int NSITES = Epi::kPaired ? NR * 2 : NR * 4, NS = NSITES / NW;
  static_assert(NS >= 1, "at least one site per wave");
  auto site_row = [&](int sidx) { return row_base + (sidx >> 2) * 16 + lk * 4 + (sidx & 3); };
  typename Epi::Pre pre[NS];
  float ln_rsum[kLnp ? NS : 1];
  float4 sq_mean[kLnp ? LNPASS : 1], sq_m2[kLnp ? LNPASS : 1];
  auto prefetch_epilogue = [&]() {
    if (nvalid > 0) {
#pragma unroll
      for (int i = 0; i < NS; ++i) pre[i] = epi.load(item, site_row(wave * NS + i), tc);
    }
    if constexpr (kLnp) {
#pragma unroll
      for (int i = 0; i < NS; ++i) ln_rsum[i] = a.ln_R[site_row(wave * NS + i)];
#pragma unroll
      for (int ps = 0; ps < LNPASS; ++ps) {
        const int c = ps * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
        const float4* p = reinterpret_cast<const float4*>(a.col_stats + ((long)item * a.T + min(t0 + min(c, COLS - 1), a.T - 1)) * 32);
        sq_mean[ps] = p[q];
        sq_m2[ps] = p[4 + q];
      }
    }
  };
  auto ln_publish = [&]() {   // after the K loop, before the reduction's barrier (convgemm.hip.h ln_publish, kColMean form)
    if constexpr (kLnp) {
#pragma unroll
      for (int ps = 0; ps < LNPASS; ++ps) {
        const int c = ps * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
        const float mg[4] = {sq_mean[ps].x, sq_mean[ps].y, sq_mean[ps].z, sq_mean[ps].w};
        const float m2g[4] = {sq_m2[ps].x, sq_m2[ps].y, sq_m2[ps].z, sq_m2[ps].w};
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
          dev2 += d * d;
        }
        dev2 += __shfl_xor(dev2, 1); dev2 += __shfl_xor(dev2, 2);
        if (q == 0 && c < COLS) {
          ln_cols[c * 2] = mean;
          ln_cols[c * 2 + 1] = 1.f / sqrtf((m2 + 32.f * dev2) / (float)(32 * a.n_groups) + a.ln_eps);
        }
      }
    }
  };

  f4 acc[NR][NM];
#pragma unroll
  for (int x = 0; x < NR; ++x)
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[x][m] = f4{0.f, 0.f, 0.f, 0.f};

  if (it_begin < it_end) {
    typedef float avec __attribute__((ext_vector_type(NR)));          // a lane's A values of one sub-step, as loaded (see RawN)
    struct Stage { avec a[2]; RawN<NM> b[2]; };                       // two K = 4 sub-steps = 8 channels
    const int n = it_end - it_begin;
    const int cb0 = it_begin / a.taps, tap0 = it_begin - cb0 * a.taps;
    constexpr unsigned ASTEP = 2u * 64u * NR * 4u;                      // bytes of A per K iteration
    const char* Abase = reinterpret_cast<const char*>(a.Wp) + ((size_t)mt * a.n_it + it_begin) * ASTEP;
    const char* Xbase = reinterpret_cast<const char*>(a.X + item * a.x_bstride + a.shift0 + t0);
    const unsigned rs = (unsigned)a.ldx * 4u;
    const unsigned d_tap = (unsigned)a.dshift * 4u;
    const unsigned d_wrap = 8u * rs - (unsigned)(a.taps - 1) * d_tap;
    const int itl = it_end - 1, cbl = itl / a.taps, tapl = itl - cbl * a.taps;
    const unsigned a_last = (unsigned)(n - 1) * ASTEP;
    const unsigned x_last = (unsigned)cbl * 8u * rs + (unsigned)tapl * d_tap;
    unsigned a_off = 0, x_off = (unsigned)cb0 * 8u * rs + (unsigned)tap0 * d_tap;
    int tap = tap0;
    const unsigned a_lane = lane * (NR * 4u);
    const unsigned x_lane0 = (unsigned)lk * rs + (unsigned)lj * (NM * 4u), x_lane1 = x_lane0 + 4u * rs;

    auto load = [&](Stage& s) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const char* pa = Abase + (a_off + a_lane + h * (64u * NR * 4u));
        s.a[h] = *reinterpret_cast<const avec*>(pa);
      }
      s.b[0] = ldRaw<NM>(reinterpret_cast<const float*>(Xbase + (x_off + x_lane0)));
      s.b[1] = ldRaw<NM>(reinterpret_cast<const float*>(Xbase + (x_off + x_lane1)));
      const bool wrap = tap + 1 == a.taps;
      a_off = min(a_off + ASTEP, a_last);
      x_off = min(x_off + (wrap ? d_wrap : d_tap), x_last);
      tap = wrap ? 0 : tap + 1;
    };
    auto compute = [&](Stage& s) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float bv[NM];
        bv[0] = s.b[h].template get<0>(); bv[1] = s.b[h].template get<1>(); bv[2] = s.b[h].template get<2>(); bv[3] = s.b[h].template get<3>();
        if constexpr (NM > 4) bv[4] = s.b[h].template get<4>();
        if constexpr (NM > 5) bv[5] = s.b[h].template get<5>();
        if constexpr (NM > 6) bv[6] = s.b[h].template get<6>();
        if constexpr (NM > 7) bv[7] = s.b[h].template get<7>();
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
          for (int x = 0; x < NR; ++x) acc[x][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s.a[h][x], bv[m], acc[x][m], 0, 0, 0);
      }
    };
    constexpr int NLOADS = 2 + (NM == 4 ? 2 : 4);                       // vector loads per slot
    constexpr int NMFMA = 2 * NM * NR;
    auto slot = [&](Stage& Ld, Stage& C) {
      load(Ld);
      compute(C);
#pragma unroll
      for (int k = 0; k < NLOADS; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMAs (2 x 32 cycles)
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // a few VALU / SALU
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - 2 * NLOADS, 0);
      __builtin_amdgcn_sched_barrier(0);
    };

    constexpr int D = 4;
    Stage st[D];
    FDX_STAMP(1);
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load(st[d]);
    __builtin_amdgcn_sched_barrier(0);
    prefetch_epilogue();
    __builtin_amdgcn_sched_barrier(0);
    int done = 0;
    for (; done + D <= n; done += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) slot(st[(d + D - 1) % D], st[d]);
#ifdef FDX_KTRACE
      if (done == 0) FDX_STAMP(6);
#endif
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
      if (done + d < n) compute(st[d]);
  } else {
    prefetch_epilogue();
  }
  FDX_STAMP(2);
  ln_publish();

  // ---- cross-wave K reduction through LDS, fixed order w0 + w1 + w2 + w3
  f4* redv = reinterpret_cast<f4*>(red);
  constexpr int Q = STR / 4;                                            // float4 slots per (row, lane)
#if defined(FDX_BISECT) && FDX_BISECT == 3
  constexpr bool kReduce = false;                                       // (clock bisect: no LDS writes, no barrier, every wave keeps its own partial sums)
#else
  constexpr bool kReduce = true;
#endif
  if constexpr (kReduce)
#pragma unroll
  for (int x = 0; x < NR; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int base = ((wave * (NR * 4) + x * 4 + r) * kWave + lane) * Q;
      float t[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) t[m] = 0.f;
#pragma unroll
      for (int m = 0; m < NM; ++m) t[m] = acc[x][m][r];
      redv[base] = f4{t[0], t[1], t[2], t[3]};
      if constexpr (NM > 4) redv[base + 1] = f4{t[4], t[5], t[6], t[7]};
    }
  FDX_STAMP(3);
  if constexpr (kReduce) __syncthreads();
  FDX_STAMP(4);
  if (nvalid <= 0) return;
  auto rsum = [&](int s) {   // s = x*4 + reg
    VecN<NM> out;
    if constexpr (!kReduce) {
#pragma unroll
      for (int m = 0; m < NM; ++m) out.v[m] = acc[(s >> 2) % NR][m][s & 3];
      return out;
    }
    f4 lo = redv[((0 * (NR * 4) + s) * kWave + lane) * Q];
    f4 hi = f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (NM > 4) hi = redv[((0 * (NR * 4) + s) * kWave + lane) * Q + 1];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      lo += redv[((w * (NR * 4) + s) * kWave + lane) * Q];
      if constexpr (NM > 4) hi += redv[((w * (NR * 4) + s) * kWave + lane) * Q + 1];
    }
    out.v[0] = lo[0]; out.v[1] = lo[1]; out.v[2] = lo[2]; out.v[3] = lo[3];
#pragma unroll
    for (int m = 4; m < NM; ++m) out.v[m] = hi[m - 4];
    return out;
  };
  float ln_mean[kLnp ? NM : 1], ln_rstd[kLnp ? NM : 1];
  if constexpr (kLnp) {
#pragma unroll
    for (int m = 0; m < NM; ++m) { ln_mean[m] = ln_cols[(NM * lj + m) * 2]; ln_rstd[m] = ln_cols[(NM * lj + m) * 2 + 1]; }
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int sidx = wave * NS + i;
    if constexpr (Epi::kPaired) epi.store(item, site_row(sidx), tc, nvalid, rsum(sidx), rsum(sidx + NR * 2), pre[i]);
    else if constexpr (kLnp) {
      VecN<NM> v = rsum(sidx);
#pragma unroll
      for (int m = 0; m < NM; ++m) v.v[m] = (v.v[m] - ln_mean[m] * ln_rsum[i]) * ln_rstd[m];
      epi.store(item, site_row(sidx), tc, nvalid, v, pre[i]);
    } else epi.store(item, site_row(sidx), tc, nvalid, rsum(sidx), pre[i]);
  }
  FDX_STAMP_END();
}

template <class Epi, int NR, int NM, int PRE = PRE_NONE>
inline hipError_t launch_convgemm16s(const ConvGeom& g, const void* Wp, const float* X, long x_bstride, int ldx, const Epi& epi,
                                     hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const float* col_stats = nullptr,
                                     const float* ln_R = nullptr, int n_groups = 0, float ln_eps = 0.f) {
  ConvArgs a;
  a.Wp = static_cast<const float4*>(Wp); a.X = X; a.x_bstride = x_bstride; a.ldx = ldx;
  a.n_it = g.cin8 * g.taps; a.taps = g.taps; a.shift0 = g.shift0; a.dshift = g.dshift;
  a.T = g.T;
  a.tiles_per_item = (g.T + 16 * NM - 1) / (16 * NM);
  a.n_tiles_n = g.B * a.tiles_per_item;
  a.n_mtiles = g.n_mtiles;             // tiles of NR 16-row blocks
  a.xcd_rect = use_xcd_rect(a.n_tiles_n, a.n_mtiles, a.taps);
  a.in_slope = 1.f;
  a.col_stats = col_stats; a.ln_R = ln_R; a.n_groups = n_groups; a.ln_eps = ln_eps;
  const int grid = conv_rect_grid(a.n_tiles_n, a.n_mtiles, a.xcd_rect);
  if (grid <= 0) return hipSuccess;
#ifdef FDX_KTRACE
  a.trace = nullptr;
  if (g_trace.buf && g_trace.n < g_trace.max_launches && grid <= g_trace.blocks_cap)
    a.trace = g_trace.buf + (size_t)(g_trace.n++) * g_trace.blocks_cap * 32;
#endif
  if (ev_start)
    hipExtLaunchKernelGGL((convgemm16s_kernel<Epi, NR, NM, PRE>), dim3(grid), dim3(256), 0, s, ev_start, ev_stop, 0, FDX_CONV_HOT_ARGS(a), conv_cold_of(a), epi);
  else
    hipLaunchKernelGGL((convgemm16s_kernel<Epi, NR, NM, PRE>), dim3(grid), dim3(256), 0, s, FDX_CONV_HOT_ARGS(a), conv_cold_of(a), epi);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ shape choice
struct Shape16 { int NR, NM; };
// rows16: 16-row blocks of the GEMM (paired: 16-pair blocks x 2).  Cost model: workgroups run in rounds of one per CU (256);
// a workgroup's time = its MFMA issue (2 * NM * NR MFMAs of 32 cycles per K iteration; the K loop is the same for every shape,
// so only the per-iteration count matters) + a fixed ~12 k cycles of set-up / pipeline fill / reduction / epilogue measured with
// tools/ktrace.py, expressed in MFMA-equivalents of this launch's K loop by the caller (fixed_per_k).  NR = 4 reads a third fewer
// operand bytes per MFMA, so it wins ties (2 % margin).
// The model deliberately does NOT credit two co-resident half-size workgroups with hiding each other's fixed phases: measured in round 3
// on the headline geometry (profiles/r03_shape_sweeps.txt), 448 workgroups of 32 x 64 or 512 of 16 x 112 (two per CU, half the MFMA work
// each) run the out-projection in 14.1 / 13.3 us against 13.3 us for 256 of 32 x 112, and the dilated conv in 28.6 against 25.5 -- the
// fixed part of a launch is a serial chain (dispatch ramp, one cold round trip in front of the first MFMA, the store drain), not idle
// issue slots a second workgroup could fill.  NR = 1 (16-row tiles, out-projection only) is therefore instantiated but never picked.
inline Shape16 pick_shape16(int rows16, int B, int T, double fixed_mfma_equiv, int n_cu = 256) {
  Shape16 best{4, 4};
  double best_cost = 1e300;
  for (int NR = 4; NR >= 2; NR -= 2) {
    if (rows16 % NR) continue;
    for (int NM = 4; NM <= 8; ++NM) {
      const long wgs = (long)(rows16 / NR) * B * ((T + 16 * NM - 1) / (16 * NM));
      const long rounds = (wgs + n_cu - 1) / n_cu;
      double cost = (double)rounds * (2.0 * NM * NR + fixed_mfma_equiv);
      if (NR == 2) cost *= 1.02;
      if (cost < best_cost - 1e-9) { best_cost = cost; best = Shape16{NR, NM}; }
    }
  }
  return best;
}

// NR = 2 fragment order from the NR = 4 one:  dst[2*mt + b][it][h][lane][half] = src[mt][it][h][lane][half*2 + b]
// (paired layout: rbk 0,1 = gate blocks, 2,3 = filter blocks of a 32-pair tile; unpaired: rbk = consecutive 16-row blocks, for
// which the same formula yields tiles {rbk 0, 2} and {1, 3} -- so unpaired weights use `pairs_mode = 0`: dst[2*mt + b][..][half] =
// src[mt][..][2*b + half]).
static __global__ void k_repack16_nr2(float2* __restrict__ dst, const float4* __restrict__ src, size_t n_src, int n_it, int paired) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // index of a source float4 = ((mt*n_it + it)*2 + h)*64 + lane
  if (i >= n_src) return;
  const float4 v = src[i];
  const size_t per_mt = (size_t)n_it * 128;
  const size_t mt = i / per_mt, rem = i - mt * per_mt;
  const float2 b0 = paired ? float2{v.x, v.z} : float2{v.x, v.y};
  const float2 b1 = paired ? float2{v.y, v.w} : float2{v.z, v.w};
  dst[(2 * mt) * per_mt + rem] = b0;
  dst[(2 * mt + 1) * per_mt + rem] = b1;
}


// 16x16x4 fragment order (NR = 4: one float4 per lane, NR = 2: one float2) of a plain [rows][K] GEMM weight, derived on the device from
// the 32x32x2 order pack_convgemm(RB = 2) wrote into the arena: src[(((mt*n_it + it)*2 + rb)*64 + hi*32 + r32)*4 + j] =
// w(row = mt*64 + rb*32 + r32, c = it*8 + hi*4 + j).  One thread per destination lane slot.
// ... and from the RB = 1 (32-row tiles) 32x32x2 order: src[((mt32 * n_it + it) * 64 + hi * 32 + r32) * 4 + j] = w(row = mt32 * 32 + r32, c = it*8 + hi*4 + j);
// n_mt32 = 32-row tiles of the source (rows = 32 n_mt32, a multiple of 16 NR)
template <int NR>
static __global__ void k_repack16_from32rb1(float* __restrict__ dst, const float* __restrict__ src, int n_mt32, int n_it) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // ((mt' * n_it + it) * 2 + h) * 64 + lane, mt' = tile of NR 16-row blocks
  const size_t n_dst = (size_t)n_mt32 * 32 / (16 * NR) * n_it * 128;
  if (i >= n_dst) return;
  const int lane = (int)(i & 63), h = (int)((i >> 6) & 1);
  const size_t q = i >> 7;
  const int it = (int)(q % n_it), mtp = (int)(q / n_it);
  const int c = it * 8 + h * 4 + (lane >> 4);
#pragma unroll
  for (int x = 0; x < NR; ++x) {
    const int row = mtp * (16 * NR) + x * 16 + (lane & 15);
    const int mt = row >> 5, r32 = row & 31, hi = (c & 7) >> 2, j = c & 3;
    dst[i * NR + x] = src[(((size_t)mt * n_it + (c >> 3)) * 64 + hi * 32 + r32) * 4 + j];
  }
}

template <int NR>
static __global__ void k_repack16_from32(float* __restrict__ dst, const float* __restrict__ src, int n_mt64, int n_it) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // ((mt'*n_it + it)*2 + h)*64 + lane, mt' = tile of NR blocks
  const size_t n_dst = (size_t)n_mt64 * (4 / NR) * n_it * 128;
  if (i >= n_dst) return;
  const int lane = (int)(i & 63), h = (int)((i >> 6) & 1);
  const size_t q = i >> 7;
  const int it = (int)(q % n_it), mtp = (int)(q / n_it);
  const int c = it * 8 + h * 4 + (lane >> 4);
#pragma unroll
  for (int x = 0; x < NR; ++x) {
    const int row = mtp * (16 * NR) + x * 16 + (lane & 15);
    const int mt = row >> 6, rb = (row >> 5) & 1, r32 = row & 31, hi = (c & 7) >> 2, j = c & 3;
    dst[i * NR + x] = src[((((size_t)mt * n_it + (c >> 3)) * 2 + rb) * 64 + hi * 32 + r32) * 4 + j];
  }
}

}  // namespace fdx
