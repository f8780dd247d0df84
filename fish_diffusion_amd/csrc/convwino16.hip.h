// convwino16.hip.h -- the residual block's dilated conv k = 3 + gate as a Winograd F(2, 3) contraction over the DILATED axis (round 6, opt-in:
// FDX_WN_WINO=1; the default path is convgemm16s.hip.h).
//
// The conv y[t] = g0 x[t-d] + g1 x[t] + g2 x[t+d] (wavenet.py:88-95, dilation d = 2^(i % cycle)) pairs the outputs (t, t + d):
//   m0 = (d0 - d2) g0     m1 = (d1 + d2) (g0 + g1 + g2) / 2     m2 = (d2 - d1) (g0 - g1 + g2) / 2     m3 = (d1 - d3) g2
//   y[t] = m0 + m1 + m2   y[t + d] = m1 - m2 - m3               with  d0 .. d3 = x[t - d], x[t], x[t + d], x[t + 2d]
// -- 4 products per 2 outputs instead of 6: two thirds of the MFMAs of the direct sum, at the operand traffic of a 2-tap loop (the timing probe
// FDX_PROBE_TAPS=2 measured that mix at 21.0 us against 25.4 for the direct kernel, profiles/r06_winograd_instruction_mix_probe.txt).  Everything
// stays fp32: the four transformed weight sets U0 .. U3 are formed once at attach (k_repack_wino), the transformed inputs per lane in the K loop
// (4 VALU ops per pair and 4-channel sub-step), products and sums on v_mfma_f32_16x16x4_f32, the output transform per wave before the fixed-order
// cross-wave reduction.  The rounding differs from the direct sum's (a few ulp of the partial products more); the result of an element does not
// depend on the tile that computed it, and pairs are anchored to the ROW (column blocks of 2 d), so an exact-ragged item is bit-identical to its
// batch-1 run when it starts at a multiple of 2 d_max (fish_diffusion_amd/diffusion.py aligns items to 32 frames in this mode).
//
// Tile = 32 pair-rows (one 16-channel gate block + its filter block: NR = 2) x 32 NP columns, NP in {2, 4} pairs per lane; 4 waves split K
// (cb = 8-channel blocks) exactly like the direct kernel; a lane owns the 2 NP columns of NP pairs as two groups of NP adjacent columns:
//   DM = 0 (d >= NP, NP | d):  pairs q0 .. q0 + NP - 1 of one d-block: evens te .. te + NP - 1, odds te + d .. ; four NP-vector loads per sub-step
//   DM = 1 (d = 1):            columns t0 .. t0 + 2 NP - 1, pairs (t0 + 2m, t0 + 2m + 1); one window x[t0 - 1 .. t0 + 2 NP]
//   DM = 2 (d = 2, NP = 4):    columns t0 .. t0 + 7, pairs (t0, t0+2) (t0+1, t0+3) (t0+4, t0+6) (t0+5, t0+7); window x[t0 - 2 .. t0 + 9]
#pragma once
#include "convgemm16s.hip.h"

namespace fdx {

// U weights from the NR = 2 direct fragment order (k_repack16_nr2: one float2 = (gate block, filter block) value per lane, index
// ((mt * n_it + cb * 3 + tap) * 2 + h) * 64 + lane) to [mt][cb][h][lane][comp][x]: eight floats per lane and 4-channel sub-step.
static __global__ void k_repack_wino(float* __restrict__ dst, const float2* __restrict__ src, size_t n_slots, int cin8) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // ((mt * cin8 + cb) * 2 + h) * 64 + lane
  if (i >= n_slots) return;
  const int lane = (int)(i & 63), h = (int)((i >> 6) & 1);
  const size_t q = i >> 7;
  const int cb = (int)(q % cin8);
  const size_t mt = q / cin8;
  float2 g[3];
#pragma unroll
  for (int tap = 0; tap < 3; ++tap) g[tap] = src[((mt * (size_t)(cin8 * 3) + cb * 3 + tap) * 2 + h) * 64 + lane];
  float* o = dst + i * 8;
  o[0] = g[0].x;                                   o[1] = g[0].y;
  o[2] = ((g[0].x + g[1].x) + g[2].x) * 0.5f;      o[3] = ((g[0].y + g[1].y) + g[2].y) * 0.5f;
  o[4] = ((g[0].x - g[1].x) + g[2].x) * 0.5f;      o[5] = ((g[0].y - g[1].y) + g[2].y) * 0.5f;
  o[6] = g[2].x;                                   o[7] = g[2].y;
}
inline size_t wino_floats(int n_mt2, int cin8) { return (size_t)n_mt2 * cin8 * 2 * 64 * 8; }

template <int NP> struct WinoVec { typedef f4a type; };
template <> struct WinoVec<2> { typedef f2a type; };

template <class Epi, int NP, int DM>
__global__ __launch_bounds__(256) void convwino16_kernel(FDX_CONV_HOT_PARAMS, ConvArgsCold cold, Epi epi) {
  FDX_CONV_ARGS_FROM_HOT(cold);
  static_assert(NP == 2 || NP == 4, "2 or 4 pairs per lane");
  static_assert(DM == 0 || DM == 1 || (DM == 2 && NP == 4), "window modes");
  constexpr int NR = 2, NW = 4, COLS = 32 * NP, NM = 2 * NP, NA = 4 * NP;   // NA: accumulators per row block (4 components x NP pairs)
  constexpr int Q = NM / 4 > 0 ? NM / 4 : 1;                                 // float4 slots of a lane's NM partial sums in LDS
  typedef typename WinoVec<NP>::type vnp;
  a.tiles_per_item = (a.T + COLS - 1) / COLS;
  __shared__ float red[NW * NR * 4 * kWave * Q * 4];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lj = lane & 15, lk = lane >> 4;
  const int d = a.dshift;                                                    // the dilation (shift0 = -d)
  FDX_STAMP(0);
  FDX_STAMP_RT0();

  int mt, nt;
  if (!conv_tile_of_block(a.n_tiles_n, a.n_mtiles, a.xcd_rect, blockIdx.x, mt, nt)) return;
  const int item = nt / a.tiles_per_item;
  const int t0 = (nt - item * a.tiles_per_item) * COLS;
  // this lane's two column groups (NP adjacent columns each)
  int tA, tB, tload;
  if constexpr (DM == 0) {
    const int q0 = NP * lj;
    const int te = 2 * d * (q0 / d) + (q0 % d);
    tA = t0 + te; tB = tA + d; tload = tA - d;
  } else {
    tA = t0 + NM * lj; tB = tA + NP; tload = tA - d;
  }
  const int nvA = min(NP, a.T - tA), nvB = min(NP, a.T - tB);
  const int row_base = mt * 16;                                              // pair rows of the tile (16 gate channels + their filter rows)

  const int per = (a.n_it + NW - 1) / NW;
  const int it_begin = wave * per, it_end = min(a.n_it, it_begin + per);

  const int my_row = row_base + lk * 4 + wave;                               // this wave's epilogue site: register `wave` of every lane's row quad
  typename Epi::Pre preA, preB;
  auto prefetch_epilogue = [&]() {
    if (nvA > 0) preA = epi.load(item, my_row, tA);
    if (nvB > 0) preB = epi.load(item, my_row, tB);
  };

  f4 acc[NR][NA];
#pragma unroll
  for (int x = 0; x < NR; ++x)
#pragma unroll
    for (int m = 0; m < NA; ++m) acc[x][m] = f4{0.f, 0.f, 0.f, 0.f};

  if (it_begin < it_end) {
    struct Raw { vnp v0, v1, v2, v3; };                                      // DM 0: x[t-d], x[t], x[t+d], x[t+2d] of the lane's NP pairs
    struct RawW { f4a w0, w1, w2; };                                         // DM 1 / 2: the window (NP = 4: 10 / 12 floats; NP = 2: 6)
    struct Stage { f4a a[2][2]; Raw b[2]; RawW w[2]; };
    const int n = it_end - it_begin;
    constexpr unsigned ASTEP = 2u * 64u * 8u * 4u;                            // bytes of U per 8-channel block
    const char* Abase = reinterpret_cast<const char*>(a.Wp) + ((size_t)mt * a.n_it + it_begin) * ASTEP;
    const char* Xbase = reinterpret_cast<const char*>(a.X + item * a.x_bstride + tload);
    const unsigned rs = (unsigned)a.ldx * 4u;
    const unsigned a_last = (unsigned)(n - 1) * ASTEP;
    const unsigned x_last = (unsigned)(it_end - 1) * 8u * rs;
    unsigned a_off = 0, x_off = (unsigned)it_begin * 8u * rs;
    const unsigned a_lane = lane * 32u;
    const unsigned x_lane0 = (unsigned)lk * rs, x_lane1 = x_lane0 + 4u * rs;
    const unsigned dB = (unsigned)d * 4u;

    auto load = [&](Stage& s) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const char* pa = Abase + (a_off + a_lane + h * (64u * 32u));
        s.a[h][0] = *reinterpret_cast<const f4a*>(pa);
        s.a[h][1] = *reinterpret_cast<const f4a*>(pa + 16);
        const char* px = Xbase + (x_off + (h ? x_lane1 : x_lane0));
        if constexpr (DM == 0) {
          s.b[h].v0 = *reinterpret_cast<const vnp*>(px);
          s.b[h].v1 = *reinterpret_cast<const vnp*>(px + dB);
          s.b[h].v2 = *reinterpret_cast<const vnp*>(px + 2u * dB);
          s.b[h].v3 = *reinterpret_cast<const vnp*>(px + 3u * dB);
        } else {
          s.w[h].w0 = *reinterpret_cast<const f4a*>(px);
          if constexpr (NP == 4) s.w[h].w1 = *reinterpret_cast<const f4a*>(px + 16);
          if constexpr (NP == 4 && DM == 2) s.w[h].w2 = *reinterpret_cast<const f4a*>(px + 32);
          if constexpr (NP == 4 && DM == 1) { const f2a t = *reinterpret_cast<const f2a*>(px + 32); s.w[h].w2 = f4a{t[0], t[1], 0.f, 0.f}; }
          if constexpr (NP == 2) { const f2a t = *reinterpret_cast<const f2a*>(px + 16); s.w[h].w1 = f4a{t[0], t[1], 0.f, 0.f}; }
        }
      }
      a_off = min(a_off + ASTEP, a_last);
      x_off = min(x_off + 8u * rs, x_last);
    };
    // one 4-channel sub-step: the lane's transformed inputs (4 VALU ops per pair), then 4 NP NR MFMAs on 4 NP NR DIFFERENT accumulators
    auto compute_h = [&](Stage& s, int h) __attribute__((always_inline)) {
      float d0[NP], d1[NP], d2[NP], d3[NP];
      if constexpr (DM == 0) {
#pragma unroll
        for (int m = 0; m < NP; ++m) { d0[m] = s.b[h].v0[m]; d1[m] = s.b[h].v1[m]; d2[m] = s.b[h].v2[m]; d3[m] = s.b[h].v3[m]; }
      } else {
        float f[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f[i] = s.w[h].w0[i];
          f[4 + i] = s.w[h].w1[i];
          if constexpr (NP == 4) f[8 + i] = s.w[h].w2[i]; else f[8 + i] = 0.f;
        }
#pragma unroll
        for (int m = 0; m < NP; ++m) {
          const int c = DM == 1 ? 2 * m : 4 * (m >> 1) + (m & 1);     // column of pair m's first output relative to the lane's first column
          d0[m] = f[c]; d1[m] = f[c + DM]; d2[m] = f[c + 2 * DM]; d3[m] = f[c + 3 * DM];     // (the window starts at column -d)
        }
      }
      float v[4][NP];
#pragma unroll
#if defined(FDX_BISECT) && FDX_BISECT == 5
      for (int m = 0; m < NP; ++m) { v[0][m] = d0[m]; v[1][m] = d1[m]; v[2][m] = d2[m]; v[3][m] = d3[m]; }     // (timing bisect: no input transform; results wrong)
#else
      for (int m = 0; m < NP; ++m) { v[0][m] = d0[m] - d2[m]; v[1][m] = d1[m] + d2[m]; v[2][m] = d2[m] - d1[m]; v[3][m] = d1[m] - d3[m]; }
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < NP; ++m)
#pragma unroll
          for (int x = 0; x < NR; ++x)
            acc[x][j * NP + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s.a[h][j >> 1][(j & 1) * 2 + x], v[j][m], acc[x][j * NP + m], 0, 0, 0);
    };
    auto compute = [&](Stage& s) {
      compute_h(s, 0);
      __builtin_amdgcn_sched_barrier(0);       // (the two sub-steps hit the SAME accumulators: left alone hipcc issues them as dependent back-to-back pairs)
      compute_h(s, 1);
    };
    constexpr int NLH = 2 + (DM == 0 ? 4 : (NP == 4 ? 3 : 2));                   // vector loads per sub-step
    constexpr int NMH = NA * NR;                                                  // MFMAs per sub-step
    auto load_h = [&](Stage& s, int h) __attribute__((always_inline)) {
      const char* pa = Abase + (a_off + a_lane + h * (64u * 32u));
      s.a[h][0] = *reinterpret_cast<const f4a*>(pa);
      s.a[h][1] = *reinterpret_cast<const f4a*>(pa + 16);
      const char* px = Xbase + (x_off + (h ? x_lane1 : x_lane0));
      if constexpr (DM == 0) {
        s.b[h].v0 = *reinterpret_cast<const vnp*>(px);
        s.b[h].v1 = *reinterpret_cast<const vnp*>(px + dB);
        s.b[h].v2 = *reinterpret_cast<const vnp*>(px + 2u * dB);
        s.b[h].v3 = *reinterpret_cast<const vnp*>(px + 3u * dB);
      } else {
        s.w[h].w0 = *reinterpret_cast<const f4a*>(px);
        if constexpr (NP == 4) s.w[h].w1 = *reinterpret_cast<const f4a*>(px + 16);
        if constexpr (NP == 4 && DM == 2) s.w[h].w2 = *reinterpret_cast<const f4a*>(px + 32);
        if constexpr (NP == 4 && DM == 1) { const f2a t = *reinterpret_cast<const f2a*>(px + 32); s.w[h].w2 = f4a{t[0], t[1], 0.f, 0.f}; }
        if constexpr (NP == 2) { const f2a t = *reinterpret_cast<const f2a*>(px + 16); s.w[h].w1 = f4a{t[0], t[1], 0.f, 0.f}; }
      }
    };
    auto slot = [&](Stage& Ld, Stage& C) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        load_h(Ld, h);
        if (h == 1) {
          a_off = min(a_off + ASTEP, a_last);
          x_off = min(x_off + 8u * rs, x_last);
        }
        compute_h(C, h);
#pragma unroll
        for (int k = 0; k < NLH; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMAs
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
          __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // a few VALU / SALU
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMH - 2 * NLH, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    constexpr int D = 3;
    Stage st[D];
    FDX_STAMP(1);
#pragma unroll
    for (int dd = 0; dd < D - 1; ++dd) load(st[dd]);
    __builtin_amdgcn_sched_barrier(0);
    prefetch_epilogue();
    __builtin_amdgcn_sched_barrier(0);
    int done = 0;
    for (; done + D <= n; done += D) {
#pragma unroll
      for (int dd = 0; dd < D; ++dd) slot(st[(dd + D - 1) % D], st[dd]);
#ifdef FDX_KTRACE
      if (done == 0) FDX_STAMP(6);
#endif
    }
#pragma unroll
    for (int dd = 0; dd < D - 1; ++dd)
      if (done + dd < n) compute(st[dd]);
  } else {
    prefetch_epilogue();
  }

  FDX_STAMP(2);
  // ---- output transform per wave, then the cross-wave K reduction through LDS in the fixed order w0 + w1 + w2 + w3
  f4* redv = reinterpret_cast<f4*>(red);
#pragma unroll
  for (int x = 0; x < NR; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y0[NP], y1[NP];
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        const float m0 = acc[x][m][r], m1 = acc[x][NP + m][r], m2 = acc[x][2 * NP + m][r], m3 = acc[x][3 * NP + m][r];
        y0[m] = (m0 + m1) + m2;
        y1[m] = (m1 - m2) - m3;
      }
      float gA[4] = {0.f, 0.f, 0.f, 0.f}, gB[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (DM == 0) {
#pragma unroll
        for (int m = 0; m < NP; ++m) { gA[m] = y0[m]; gB[m] = y1[m]; }
      } else if constexpr (DM == 1 && NP == 4) {
        gA[0] = y0[0]; gA[1] = y1[0]; gA[2] = y0[1]; gA[3] = y1[1];
        gB[0] = y0[2]; gB[1] = y1[2]; gB[2] = y0[3]; gB[3] = y1[3];
      } else if constexpr (DM == 1 && NP == 2) {
        gA[0] = y0[0]; gA[1] = y1[0]; gB[0] = y0[1]; gB[1] = y1[1];
      } else {
        gA[0] = y0[0]; gA[1] = y0[1]; gA[2] = y1[0]; gA[3] = y1[1];
        gB[0] = y0[2]; gB[1] = y0[3]; gB[2] = y1[2]; gB[3] = y1[3];
      }
      const int base = ((wave * (NR * 4) + x * 4 + r) * kWave + lane) * Q;
      if constexpr (NP == 4) { redv[base] = f4{gA[0], gA[1], gA[2], gA[3]}; redv[base + 1] = f4{gB[0], gB[1], gB[2], gB[3]}; }
      else redv[base] = f4{gA[0], gA[1], gB[0], gB[1]};
    }
  FDX_STAMP(3);
  __syncthreads();
  FDX_STAMP(4);
  if (nvA <= 0) return;                                                      // (tA < tB: nothing of this lane exists)
  auto rsum = [&](int s, VecN<NP>& oA, VecN<NP>& oB) {                        // s = x * 4 + reg
    f4 lo = redv[((0 * (NR * 4) + s) * kWave + lane) * Q];
    f4 hi = f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (NP == 4) hi = redv[((0 * (NR * 4) + s) * kWave + lane) * Q + 1];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      lo += redv[((w * (NR * 4) + s) * kWave + lane) * Q];
      if constexpr (NP == 4) hi += redv[((w * (NR * 4) + s) * kWave + lane) * Q + 1];
    }
    if constexpr (NP == 4) {
#pragma unroll
      for (int m = 0; m < 4; ++m) { oA.v[m] = lo[m]; oB.v[m] = hi[m]; }
    } else {
      oA.v[0] = lo[0]; oA.v[1] = lo[1]; oB.v[0] = lo[2]; oB.v[1] = lo[3];
    }
  };
  VecN<NP> gA, gB, fA, fB;
  rsum(wave, gA, gB);              // gate block (x = 0), register `wave`
  rsum(4 + wave, fA, fB);          // filter block (x = 1)
  epi.store(item, my_row, tA, nvA, gA, fA, preA);
  if (nvB > 0) epi.store(item, my_row, tB, nvB, gB, fB, preB);
  FDX_STAMP_END();
}

template <class Epi, int NP, int DM>
inline hipError_t launch_convwino16_one(ConvArgs& a, int grid, const Epi& epi, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (ev_start)
    hipExtLaunchKernelGGL((convwino16_kernel<Epi, NP, DM>), dim3(grid), dim3(256), 0, s, ev_start, ev_stop, 0, FDX_CONV_HOT_ARGS(a), conv_cold_of(a), epi);
  else
    hipLaunchKernelGGL((convwino16_kernel<Epi, NP, DM>), dim3(grid), dim3(256), 0, s, FDX_CONV_HOT_ARGS(a), conv_cold_of(a), epi);
  return hipGetLastError();
}

// g.n_mtiles = tiles of 32 rows (NR = 2), g.cin8 = 8-channel blocks, g.dshift = the dilation (a power of two; NP = 4 needs d in {1, 2} or 4 | d)
template <template <int> class EpiT, int NP>
inline hipError_t launch_convwino16(const ConvGeom& g, const void* U, const float* X, long x_bstride, int ldx, const EpiT<NP>& epi, hipStream_t s,
                                    hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  ConvArgs a;
  a.Wp = static_cast<const float4*>(U); a.X = X; a.x_bstride = x_bstride; a.ldx = ldx;
  a.n_it = g.cin8; a.taps = 1; a.shift0 = -g.dshift; a.dshift = g.dshift;
  a.T = g.T;
  a.tiles_per_item = (g.T + 32 * NP - 1) / (32 * NP);
  a.n_tiles_n = g.B * a.tiles_per_item;
  a.n_mtiles = g.n_mtiles;
  a.xcd_rect = use_xcd_rect(a.n_tiles_n, a.n_mtiles, 3);
  a.in_slope = 1.f;
  a.col_stats = nullptr; a.ln_R = nullptr; a.n_groups = 0; a.ln_eps = 0.f;
  const int grid = conv_rect_grid(a.n_tiles_n, a.n_mtiles, a.xcd_rect);
  if (grid <= 0) return hipSuccess;
#ifdef FDX_KTRACE
  a.trace = nullptr;
  if (g_trace.buf && g_trace.n < g_trace.max_launches && grid <= g_trace.blocks_cap)
    a.trace = g_trace.buf + (size_t)(g_trace.n++) * g_trace.blocks_cap * 32;
#endif
  const int d = g.dshift;
  if (d == 1) return launch_convwino16_one<EpiT<NP>, NP, 1>(a, grid, epi, s, ev_start, ev_stop);
  if constexpr (NP == 4) {
    if (d == 2) return launch_convwino16_one<EpiT<NP>, NP, 2>(a, grid, epi, s, ev_start, ev_stop);
  }
  if (d % NP != 0) return hipErrorInvalidValue;
  return launch_convwino16_one<EpiT<NP>, NP, 0>(a, grid, epi, s, ev_start, ev_stop);
}

}  // namespace fdx
