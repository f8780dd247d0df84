// bf16lds.hip.h -- the two residual-block GEMMs of the opt-in bf16 storage mode, LDS-tiled, for large column counts (BASELINE
// configs[4]: 16 utterances per GPU => N = 13 776 columns).
//
// The register-direct kernels (convgemm_kernel<..., OPK_BF16>) stream 4 KB of operands per 4 MFMAs per wave: with the bf16 MFMA 16x
// faster than the fp32 one that is 128 B/cycle/CU of L1/L2 traffic, and the kernel sits at ~20 % of the 2.5 PFLOP/s roof.  Here a
// workgroup owns a 128-row x 128-column tile; its operands are staged through LDS ONCE per 32-channel block and shared by the 4 waves
// (2 x 2, each 64 x 64 = 2 x 2 MFMA blocks of v_mfma_f32_32x32x16_bf16):
//
//   A (weights)      packed on the device at attach time as [m-tile][block][tap][k16-step][k-group][128 rows] x 16 B: one block is a
//                    contiguous 8 KB x taps slab, copied linearly global -> LDS; a lane's fragment (row i, 8 consecutive k) is one
//                    conflict-free ds_read_b128.
//   B (activations)  C8-blocked bf16 (16-byte group = 8 consecutive channels of one column = a lane's B fragment): per block the
//                    4 channel-group rows x (128 + 16) columns window is staged once and the conv's THREE TAPS read it at shifted
//                    columns -- the dilated-conv window in LDS that north_star asks for; a third of the activation traffic.
//   double-buffered LDS (2 x 33 KB => 2 workgroups per CU), one barrier per block; the global loads of block k+1 are in flight
//   while block k's 24 MFMAs per wave run.
//
// Paired rows (dilated conv + gate): rows 0..63 of a wave pair = gate rows of 32 channels (wave row wr) + the matching filter rows, so
// sigmoid(g) * tanh(f) is formed in registers.  Epilogues write the next GEMM's operand directly in the blocked bf16 layout.
#pragma once
#include "convgemm.hip.h"
#include <cstdlib>

namespace fdx {

struct BfArgs {
  const uint4* Wp;          // packed A of this layer: [m_tile][blk][TAPS][2][2][128] uint4
  const uint4* Xb;          // blocked bf16 activations, group (cb8, t) of item b at Xb[b*x_bs + cb8*ld + t]  (t relative to the first valid column)
  long x_bs;                // 16-byte groups between items
  int ld;                   // groups per channel-group row
  int n_blk;                // C / 32
  int dil;                  // dilation (conv) -- the window carries an 8-column halo either side
  int T, tiles_per_item, n_tiles_n, n_mtiles;
#ifdef FDX_KTRACE
  unsigned long long* trace;
#endif
};

constexpr int kBfWin = 128 + 16;   // staged columns per block: tile + 8 either side (dilation <= 8)

// ---- epilogues: called once per (row block pair | row block, column block) with the wave's accumulators
struct BfEpiGate {     // wavenet.py:112-115; Z out as blocked bf16 (the out-projection's operand)
  static constexpr int kTaps = 3;
  static constexpr bool kPaired = true;
  const float* P; long p_bs; int ldp;      // conditioner slab (+ biases), fp32 [B][2C][ldp]
  __bf16* Zb; long zb_bs; int ldz;         // blocked bf16 out: element (c, t) at ((c >> 3) * ldz + t) * 8 + (c & 7)
  int C;
};
struct BfEpiResSkip {  // wavenet.py:117-120 + the skip sum of :228; Y = x + step out as blocked bf16 (the next conv's operand)
  static constexpr int kTaps = 1;
  static constexpr bool kPaired = false;
  float* X; float* SK; long bs; int ld;    // fp32 residual stream / skip sum [B][C][ld]
  const float* bias;                       // [2C]
  const float* sb; int sb_ld, sb_bs;       // next layer's diffusion projection
  __bf16* Yb; long yb_bs;                  // blocked bf16 out (null on the last layer)
  int C, skip_mode;
  float inv_div, r_inv_div;
};

template <class Epi, int DBG = 0>
__global__ __launch_bounds__(256, 2) void bf16lds_kernel(BfArgs a, Epi epi) {
  constexpr int TAPS = Epi::kTaps;
  constexpr int A_G = TAPS * 2 * 2 * 128;           // 16-byte groups of A per block
  constexpr int B_G = 4 * kBfWin;                   // ... of B
  constexpr int A_LD = A_G / 256;                   // A loads per thread per block
  constexpr int B_LD = (B_G + 255) / 256;
  constexpr int B_GP = B_LD * 256;                  // B region padded to whole passes: every thread loads and stores unconditionally
  __shared__ uint4 lds[2][A_G + B_GP];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 5, i = lane & 31;

  // ---- tile -> XCD map: 2 row groups x 4 column groups when the counts divide (each XCD keeps its weight group L2-resident and
  // streams a quarter of the activations), else row runs; within an XCD the row tiles of one column tile run back to back
  const int G = a.n_tiles_n * a.n_mtiles, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
  int mt, nt;
  if ((a.n_mtiles & 1) == 0 && (a.n_tiles_n & 3) == 0 && (G & 7) == 0) {
    const int MH = a.n_mtiles >> 1, NQ = a.n_tiles_n >> 2;
    const int ntl = slot / MH;
    mt = (xcd & 1) * MH + (slot - ntl * MH);
    nt = (xcd >> 1) * NQ + ntl;
  } else {
    const int q8 = G >> 3, r8 = G & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    mt = L / a.n_tiles_n;
    nt = L - mt * a.n_tiles_n;
  }
  const int item = nt / a.tiles_per_item;
  const int t0 = (nt - item * a.tiles_per_item) * 128;

  const uint4* Ag = a.Wp + (size_t)((DBG & 16) ? 0 : mt) * a.n_blk * A_G;
  const uint4* Bg = (DBG & 16) ? a.Xb - 8 : a.Xb + item * a.x_bs + (t0 - 8);

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][nb][r] = 0.f;

  // Staging registers: two sets of named scalars (hipcc routes register ARRAYS filled by loads through scratch memory here, and
  // conditional loads serialise load -> wait -> ds_write: 7 k cycles per block instead of the ~1 k the MFMAs need).  Every thread
  // loads and stores unconditionally; a thread past the end of the B window re-loads its last group into the padded LDS tail.
  // Two sets = a prefetch distance of two blocks: one block's MFMAs (~0.8-1.5 k cycles) do not cover an HBM miss.
  static_assert((A_LD == 6 || A_LD == 2) && B_LD == 3, "staging code below is written out for these counts");
#define BF_DECL(S) uint4 ra0##S, ra1##S, ra2##S = uint4{0u, 0u, 0u, 0u}, ra3##S = ra2##S, ra4##S = ra2##S, ra5##S = ra2##S, rb0##S, rb1##S, rb2##S
  BF_DECL(P);
  BF_DECL(Q);
  auto boff = [&](int k) {
    const int e = min(k * 256 + tid, B_G - 1);
    const int row = e / kBfWin, col = e - row * kBfWin;
    return (size_t)row * a.ld + col;
  };
  const size_t bo0 = boff(0), bo1 = boff(1), bo2 = boff(2);
  const size_t b_blk = (size_t)4 * a.ld;
  const int last = a.n_blk - 1;
#define BF_GLOAD(S, blk_)                                                 \
  do {                                                                    \
    const int bk_ = min((blk_), last);   /* past the end: re-load the last block (unconditional, unused) */ \
    const uint4* pa_ = Ag + (size_t)bk_ * A_G + tid;                      \
    ra0##S = pa_[0]; ra1##S = pa_[256];                                   \
    if constexpr (A_LD == 6) { ra2##S = pa_[512]; ra3##S = pa_[768]; ra4##S = pa_[1024]; ra5##S = pa_[1280]; } \
    const uint4* pb_ = Bg + (size_t)bk_ * b_blk;                          \
    rb0##S = pb_[bo0]; rb1##S = pb_[bo1]; rb2##S = pb_[bo2];              \
  } while (0)
#define BF_LSTORE(S, buf_)                                                \
  do {                                                                    \
    uint4* l_ = lds[buf_] + tid;                                          \
    l_[0] = ra0##S; l_[256] = ra1##S;                                     \
    if constexpr (A_LD == 6) { l_[512] = ra2##S; l_[768] = ra3##S; l_[1024] = ra4##S; l_[1280] = ra5##S; } \
    l_[A_G] = rb0##S; l_[A_G + 256] = rb1##S; l_[A_G + 512] = rb2##S;     \
  } while (0)
  auto compute = [&](int buf) {
    const uint4* la = lds[buf];
    const uint4* lb = lds[buf] + A_G;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int shift = TAPS == 3 ? 8 + (tap - 1) * a.dil : 8;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const uint4 v = la[((tap * 2 + s) * 2 + g) * 128 + wr * 64 + x * 32 + i];
          fa[x] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const uint4 v = lb[(2 * s + g) * kBfWin + shift + wc * 64 + nb * 32 + i];
          fb[nb] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[x], fb[nb], acc[x][nb], 0, 0, 0);
      }
    }
  };

  FDX_STAMP(0);
  BF_GLOAD(P, 0);
  BF_LSTORE(P, 0);
  BF_GLOAD(Q, 1);
  __syncthreads();
  FDX_STAMP(1);
  for (int blk = 0; blk < a.n_blk; blk += 2) {
    // even block: in LDS buffer 0; set Q holds block blk+1 (in flight since the last half-iteration); block blk+2 -> set P
    BF_GLOAD(P, blk + 2);
    __builtin_amdgcn_sched_barrier(0);              // (left alone, hipcc sinks the loads below the MFMAs, right in front of their use)
    if (!(DBG & 4)) compute(0);
    __builtin_amdgcn_sched_barrier(0);
    BF_LSTORE(Q, 1);
    __syncthreads();
#ifdef FDX_KTRACE
    if (blk == 2) FDX_STAMP(3);
#endif
    // odd block: in LDS buffer 1; block blk+3 -> set Q
    BF_GLOAD(Q, blk + 3);
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 4) && blk + 1 < a.n_blk) compute(1);
    __builtin_amdgcn_sched_barrier(0);
    BF_LSTORE(P, 0);
    __syncthreads();
#ifdef FDX_KTRACE
    if (blk == 2) FDX_STAMP(4);
#endif
  }
  FDX_STAMP(2);
#undef BF_DECL
#undef BF_GLOAD
#undef BF_LSTORE

  // ---------------------------------------------------------------- epilogue
  if (DBG & 8) return;
  const int half = g;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int t = t0 + wc * 64 + nb * 32 + i;
    if (t >= a.T) continue;
    if constexpr (Epi::kPaired) {
      const int ch0 = mt * 64 + wr * 32;                       // this wave's 32 channels: gate rows = acc[0], filter rows = acc[1]
      const float* Pg = epi.P + item * epi.p_bs + t;
      float pg[16], pf[16];                                      // all 32 loads in flight before the first use
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = ch0 + acc_row(r, half);
        pg[r] = Pg[(long)ch * epi.ldp];
        pf[r] = Pg[(long)(ch + epi.C) * epi.ldp];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x4 z;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = q * 4 + k;
          z[k] = (__bf16)EpiGate::gate1(acc[0][nb][r] + pg[r], acc[1][nb][r] + pf[r]);
        }
        const int c0 = ch0 + 8 * q + 4 * half;                   // the quad's first channel (acc_row(4q, half))
        *reinterpret_cast<bf16x4*>(epi.Zb + item * epi.zb_bs + ((long)(c0 >> 3) * epi.ldz + t) * 8 + (c0 & 7)) = z;
      }
    } else {
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int row0 = mt * 128 + wr * 64 + x * 32;           // 32 rows entirely on one side of C
        const bool res = row0 < epi.C;
        // the read-modify-write streams (X, SK) may alias as far as the compiler knows: done one element at a time the 16 loads
        // serialise behind the 16 stores (107 k cycles per tile measured); all loads first, then all stores
        const long o0 = item * epi.bs + (long)(res ? row0 : row0 - epi.C) * epi.ld + t;
        float* __restrict__ RW = res ? epi.X : epi.SK;
        const bool rd = res || epi.skip_mode == 1 || epi.skip_mode == 2;
        const bool use_sb = res && epi.Yb;                      // (unconditional loads from a selected pointer: see the staging note)
        const float* sbp = use_sb ? epi.sb + item * epi.sb_bs : epi.bias;
        const long sbs = use_sb ? epi.sb_ld : 1;
        float old[16], bi[16], sbv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = acc_row(r, half);
          bi[r] = epi.bias[row0 + rr];
          old[r] = RW[o0 + (long)rr * epi.ld];                   // (read but unused when the skip sum starts here)
          sbv[r] = sbp[(long)(row0 + rr) * sbs];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bf16x4 y;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = q * 4 + k;
            float v = acc[x][nb][r] + bi[r];
            if (res) {
              v = div_const(old[r] + v, 1.41421356237309504880f, 0.70710678118654752440f);
              y[k] = (__bf16)(epi.Yb ? v + sbv[r] : 0.f);
            } else {
              if (rd) v = old[r] + v;
              if (epi.skip_mode >= 2) v = div_const(v, epi.inv_div, epi.r_inv_div);
            }
            RW[o0 + (long)acc_row(r, half) * epi.ld] = v;
          }
          if (res && epi.Yb) {
            const int c0 = row0 + 8 * q + 4 * half;
            *reinterpret_cast<bf16x4*>(epi.Yb + item * epi.yb_bs + ((long)(c0 >> 3) * epi.ld + t) * 8 + (c0 & 7)) = y;
          }
        }
      }
    }
  }
  FDX_STAMP(5);
}

template <class Epi>
inline hipError_t launch_bf16lds(const uint4* Wp, const uint4* Xb, long x_bs, int ld, int C, int dil, int B, int T, int rows, const Epi& epi,
                                 hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
  BfArgs a;
  a.Wp = Wp; a.Xb = Xb; a.x_bs = x_bs; a.ld = ld; a.n_blk = C / 32; a.dil = dil; a.T = T;
  a.tiles_per_item = (T + 127) / 128;
  a.n_tiles_n = B * a.tiles_per_item;
  a.n_mtiles = rows / 128;
  const int grid = a.n_tiles_n * a.n_mtiles;
  if (grid <= 0) return hipSuccess;
#ifdef FDX_KTRACE
  a.trace = nullptr;
  if (g_trace.buf && g_trace.n < g_trace.max_launches && grid <= g_trace.blocks_cap)
    a.trace = g_trace.buf + (size_t)(g_trace.n++) * g_trace.blocks_cap * 32;
#endif
#ifdef FDX_KTRACE
  static const int dbg = [] { const char* e = getenv("FDX_BF16_DBG"); return e ? atoi(e) : 0; }();
#define FDX_BF_DBG(D) if (dbg == D) { hipLaunchKernelGGL((bf16lds_kernel<Epi, D>), dim3(grid), dim3(256), 0, s, a, epi); return hipGetLastError(); }
  FDX_BF_DBG(1) FDX_BF_DBG(2) FDX_BF_DBG(3) FDX_BF_DBG(4) FDX_BF_DBG(7) FDX_BF_DBG(8) FDX_BF_DBG(11) FDX_BF_DBG(12) FDX_BF_DBG(16) FDX_BF_DBG(20) FDX_BF_DBG(28)
#undef FDX_BF_DBG
#endif
  if (ev0) hipExtLaunchKernelGGL((bf16lds_kernel<Epi>), dim3(grid), dim3(256), 0, s, ev0, ev1, 0, a, epi);
  else hipLaunchKernelGGL((bf16lds_kernel<Epi>), dim3(grid), dim3(256), 0, s, a, epi);
  return hipGetLastError();
}

// The LDS kernel's A order from the register-direct bf16 order (16-byte groups are moved whole):
//   new[((mt*n_blk + blk)*TAPS + tap)*2 + s)*2 + g)*128 + wr*64 + x*32 + i]
//     = old[((mt32or64*n_it + it)*2 + x)*64 + g*32 + i],   it = (blk*2 + s)*TAPS + tap,
//   paired (conv): old m-tile = 2*mt + wr (32 channels each: x = 0 gate, 1 filter);  plain (out-projection): old m-tile = 2*mt + wr (64 rows: x = 0 / 1).
static __global__ void k_bf16lds_repack(uint4* __restrict__ dst, const uint4* __restrict__ src, int n_mt, int n_blk, int taps) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per_blk = (size_t)taps * 512, total = (size_t)n_mt * n_blk * per_blk;
  if (idx >= total) return;
  const int rho = (int)(idx & 127);
  size_t q = idx >> 7;
  const int g = (int)(q & 1); q >>= 1;
  const int s = (int)(q & 1); q >>= 1;
  const int tap = (int)(q % taps); q /= taps;
  const int blk = (int)(q % n_blk);
  const int mt = (int)(q / n_blk);
  const int wr = rho >> 6, x = (rho >> 5) & 1, i = rho & 31;
  const int n_it = n_blk * 2 * taps, it = (blk * 2 + s) * taps + tap;
  dst[idx] = src[(((size_t)(2 * mt + wr) * n_it + it) * 2 + x) * 64 + g * 32 + i];
}

}  // namespace fdx
