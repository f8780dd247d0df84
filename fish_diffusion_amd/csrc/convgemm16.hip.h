// convgemm16.hip.h -- the residual-block pair of the denoiser (dilated conv + gate, out-projection + residual/skip) on
// `v_mfma_f32_16x16x4_f32` instead of 32x32x2.  Same contraction, same split-K workgroup, same epilogue arithmetic as
// convgemm.hip.h -- what changes is the operand geometry:
//
//   16x16x4 takes B as 4 k-rows x 16 columns (lane l: row l>>4, column l&15).  Let lane l load the FOUR ADJACENT columns
//   t0 + 4*(l&15) .. +3 of row l>>4 with one 16-byte load and feed value m of it to the MFMA of "column set" m: the 16x16
//   blocks then cover the interleaved column sets {4j+m}, and one dwordx4 per lane feeds 4 k-rows x 64 columns.
//   A is 16 rows x 4 k per MFMA (lane l: row l&15, k l>>4): the four 16-row blocks of the 64-row tile are packed into one
//   float4 per lane on the host.
//
//   => per 8 channels (one pipeline slot, 32 MFMAs, 1024 cycles): 2 + 2 dwordx4 loads instead of 2 dwordx4 + 4 dwordx2,
//      and every accumulator row is 4 adjacent columns: the epilogue moves 16-byte quads.
//   Measured motivation: with 32x32x2, halving the number of activation loads at equal bytes takes
//   5.5 % off the K loop -- the matrix pipe is sensitive to how many vector loads are in flight, not to their bytes.
//
// Accumulator block (rbk, m): rows rbk*16 + (l>>4)*4 + reg (reg 0..3), column t0 + 4*(l&15) + m.
// Paired layout (gate/filter): rbk 0,1 = gate rows 0..31 of the m-tile's 32 channels, rbk 2,3 = the matching filter rows.
#pragma once
#include "convgemm.hip.h"

namespace fdx {

typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };   // dword-aligned quad: global ld/st at any float offset

__device__ __forceinline__ f4 ld4(const float* p) {   // sources are the library's own padded rows (see ld2)
  const f4u v = *reinterpret_cast<const f4u*>(p);
  return f4{v.x, v.y, v.z, v.w};
}
// quad store into a padded row: columns >= T (at most 3, inside the right halo) receive 0, which is what the halo holds
__device__ __forceinline__ void st4p(float* p, f4 v, int nvalid) {
  f4u u;
  u.x = v.x; u.y = nvalid > 1 ? v.y : 0.f; u.z = nvalid > 2 ? v.z : 0.f; u.w = nvalid > 3 ? v.w : 0.f;
  // non-temporal: the tile is read next by OTHER XCDs (through the fabric, not this L2) -- see st2p in convgemm.hip.h
  typedef float f4nt __attribute__((ext_vector_type(4), aligned(4)));
  __builtin_nontemporal_store(f4nt{u.x, u.y, u.z, u.w}, reinterpret_cast<f4nt*>(p));
}

struct EpiGate16 {  // wavenet.py:112-115
  static constexpr bool kPaired = true;
  float* out; long o_bs; int ldo;
  const float* P; long p_bs; int ldp;
  int C;
  struct Pre { f4 pg, pf; };
  __device__ __forceinline__ Pre load(int b, int row, int t) const {
    const float* q = P + b * p_bs + t;
    return Pre{ld4(q + (long)row * ldp), ld4(q + (long)(row + C) * ldp)};
  }
  __device__ __forceinline__ void store(int b, int row, int t, int nvalid, f4 g, f4 f, const Pre& p) const {
    g += p.pg; f += p.pf;
    st4p(out + b * o_bs + (long)row * ldo + t,
         f4{EpiGate::gate1(g.x, f.x), EpiGate::gate1(g.y, f.y), EpiGate::gate1(g.z, f.z), EpiGate::gate1(g.w, f.w)}, nvalid);
  }
};

// ------------------------------------------------------------------------------------------ kernel
template <class Epi>
__global__ __launch_bounds__(256) void convgemm16_kernel(FDX_CONV_HOT_PARAMS, ConvArgsCold cold, Epi epi) {
  FDX_CONV_ARGS_FROM_HOT(cold);
  a.tiles_per_item = (a.T + 63) / 64;
  constexpr int NW = 4;
  __shared__ float red[NW * 16 * kWave * 4];          // [wave][rbk*4 + reg][lane][m]   (64 KB)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lj = lane & 15, lk = lane >> 4;           // B: column group / k-row;  A: row-in-block / k;  D: column group / row quad
  FDX_STAMP(0);
  FDX_STAMP_RT0();

  int mt, nt;
  if (!conv_tile_of_block(a.n_tiles_n, a.n_mtiles, a.xcd_rect, blockIdx.x, mt, nt)) return;   // (padding of an uneven rectangle map)
  const int item = nt / a.tiles_per_item;
  const int t0 = (nt - item * a.tiles_per_item) * 64;
  const int tc = t0 + 4 * lj;                         // this lane's column quad
  const int nvalid = min(4, a.T - tc);                // <= 0: the whole quad is overhang
  const int row_base = mt * (Epi::kPaired ? 32 : 64);

  const int per = (a.n_it + NW - 1) / NW;
  const int it_begin = wave * per, it_end = min(a.n_it, it_begin + per);

  // epilogue sites of this wave: paired: (rbk in {0,1}, reg) -> 8 per tile, 2 per wave; unpaired: (rbk, reg) -> 16, 4 per wave
  constexpr int NS = Epi::kPaired ? 2 : 4;
  auto site_row = [&](int sidx) {                     // sidx = rbk*4 + reg
    return row_base + (sidx >> 2) * 16 + lk * 4 + (sidx & 3);
  };
  typename Epi::Pre pre[NS];
  auto prefetch_epilogue = [&]() {
    if (nvalid > 0) {
#pragma unroll
      for (int i = 0; i < NS; ++i) pre[i] = epi.load(item, site_row(wave * NS + i), tc);
    }
  };

  f4 acc[4][4];                                        // [rbk][m]
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[x][m] = f4{0.f, 0.f, 0.f, 0.f};

  if (it_begin < it_end) {
    struct Stage { float4 a[2]; f4 b[2]; };            // two K=4 sub-steps = 8 channels
    const int n = it_end - it_begin;
    const int cb0 = it_begin / a.taps, tap0 = it_begin - cb0 * a.taps;
    const char* Abase = reinterpret_cast<const char*>(a.Wp + ((long)mt * a.n_it + it_begin) * 128);
    const char* Xbase = reinterpret_cast<const char*>(a.X + item * a.x_bstride + a.shift0 + t0);
    const unsigned rs = (unsigned)a.ldx * 4u;
    const unsigned d_tap = (unsigned)a.dshift * 4u;
    const unsigned d_wrap = 8u * rs - (unsigned)(a.taps - 1) * d_tap;
    const int itl = it_end - 1, cbl = itl / a.taps, tapl = itl - cbl * a.taps;
    const unsigned a_last = (unsigned)(n - 1) * 2048u;
    const unsigned x_last = (unsigned)cbl * 8u * rs + (unsigned)tapl * d_tap;
    unsigned a_off = 0, x_off = (unsigned)cb0 * 8u * rs + (unsigned)tap0 * d_tap;
    int tap = tap0;
    const unsigned a_lane = lane * 16u;
    const unsigned x_lane0 = (unsigned)lk * rs + (unsigned)lj * 16u, x_lane1 = x_lane0 + 4u * rs;

    auto load = [&](Stage& s) {
      s.a[0] = *reinterpret_cast<const float4*>(Abase + (a_off + a_lane));
      s.a[1] = *reinterpret_cast<const float4*>(Abase + (a_off + a_lane + 1024u));
      s.b[0] = ld4(reinterpret_cast<const float*>(Xbase + (x_off + x_lane0)));
      s.b[1] = ld4(reinterpret_cast<const float*>(Xbase + (x_off + x_lane1)));
      const bool wrap = tap + 1 == a.taps;
      a_off = min(a_off + 2048u, a_last);
      x_off = min(x_off + (wrap ? d_wrap : d_tap), x_last);
      tap = wrap ? 0 : tap + 1;
    };
    auto compute = [&](Stage& s) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float av = x == 0 ? s.a[h].x : x == 1 ? s.a[h].y : x == 2 ? s.a[h].z : s.a[h].w;
            acc[x][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s.b[h][m], acc[x][m], 0, 0, 0);
          }
        }
      }
    };
    auto slot = [&](Stage& Ld, Stage& C) {
      load(Ld);
      compute(C);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMAs (2 x 32 cycles)
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // a few VALU / SALU
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 32 - 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    };

    constexpr int D = 4;
    Stage st[D];
    FDX_STAMP(1);
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load(st[d]);
    __builtin_amdgcn_sched_barrier(0);
    int done = 0;
    prefetch_epilogue();
    __builtin_amdgcn_sched_barrier(0);
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

  // ---- cross-wave K reduction through LDS, fixed order w0 + w1 + w2 + w3
  f4* redv = reinterpret_cast<f4*>(red);
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      redv[(wave * 16 + x * 4 + r) * kWave + lane] = f4{acc[x][0][r], acc[x][1][r], acc[x][2][r], acc[x][3][r]};
  FDX_STAMP(3);
  __syncthreads();
  FDX_STAMP(4);
  if (nvalid <= 0) return;
  auto rsum = [&](int s) {
    f4 v = redv[(0 * 16 + s) * kWave + lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += redv[(w * 16 + s) * kWave + lane];
    return v;
  };
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int sidx = wave * NS + i;
    if constexpr (Epi::kPaired) epi.store(item, site_row(sidx), tc, nvalid, rsum(sidx), rsum(sidx + 8), pre[i]);
    else epi.store(item, site_row(sidx), tc, nvalid, rsum(sidx), pre[i]);
  }
  FDX_STAMP_END();
}

template <class Epi>
inline hipError_t launch_convgemm16(const ConvGeom& g, const float4* Wp, const float* X, long x_bstride, int ldx, const Epi& epi,
                                    hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  ConvArgs a;
  a.Wp = Wp; a.X = X; a.x_bstride = x_bstride; a.ldx = ldx;
  a.n_it = g.cin8 * g.taps; a.taps = g.taps; a.shift0 = g.shift0; a.dshift = g.dshift;
  a.T = g.T;
  a.tiles_per_item = (g.T + 63) / 64;
  a.n_tiles_n = g.B * a.tiles_per_item;
  a.n_mtiles = g.n_mtiles;
  a.xcd_rect = use_xcd_rect(a.n_tiles_n, a.n_mtiles, a.taps);
  a.in_slope = 1.f;
  a.col_stats = nullptr; a.ln_R = nullptr; a.n_groups = 0; a.ln_eps = 0.f;
  const int grid = conv_rect_grid(a.n_tiles_n, a.n_mtiles, a.xcd_rect);
  if (grid <= 0) return hipSuccess;
#ifdef FDX_KTRACE
  a.trace = nullptr;
  if (g_trace.buf && g_trace.n < g_trace.max_launches && grid <= g_trace.blocks_cap)
    a.trace = g_trace.buf + (size_t)(g_trace.n++) * g_trace.blocks_cap * 32;
#endif
  if (ev_start)
    hipExtLaunchKernelGGL((convgemm16_kernel<Epi>), dim3(grid), dim3(256), 0, s, ev_start, ev_stop, 0, FDX_CONV_HOT_ARGS(a), conv_cold_of(a), epi);
  else
    hipLaunchKernelGGL((convgemm16_kernel<Epi>), dim3(grid), dim3(256), 0, s, FDX_CONV_HOT_ARGS(a), conv_cold_of(a), epi);
  return hipGetLastError();
}

// dst[((mt*n_it + cb*taps + tap)*2 + h)*64 + lane] (float4) = { w(mt, rbk, i = lane&15, c = cb*8 + h*4 + (lane>>4), tap) : rbk = 0..3 }
template <class F>
inline void pack_convgemm16(float* dst, int n_mtiles, int cin8, int taps, F getw) {
  const int n_it = cin8 * taps;
  for (int mt = 0; mt < n_mtiles; ++mt)
    for (int cb = 0; cb < cin8; ++cb)
      for (int tap = 0; tap < taps; ++tap)
        for (int h = 0; h < 2; ++h) {
          float* d = dst + ((((size_t)mt * n_it + (size_t)cb * taps + tap) * 2 + h) * 64) * 4;
          for (int lane = 0; lane < 64; ++lane)
            for (int rbk = 0; rbk < 4; ++rbk)
              d[lane * 4 + rbk] = getw(mt, rbk, lane & 15, cb * 8 + h * 4 + (lane >> 4), tap);
        }
}

}  // namespace fdx
