// bf16lds.hip.h -- the two residual-block GEMMs of the two opt-in 16-bit-MFMA modes, LDS-tiled, for large column counts (BASELINE
// configs[4]: 16 utterances per GPU => N = 13 776 columns):  the bf16 storage mode (F16S = 0; not parity-grade) and the fp16-split
// mode (F16S = 1; fp32-class: operands hi + lo in fp16, three MFMAs per product block -- see the kernel's comment).
//
// The register-direct kernels (convgemm_kernel<..., OPK_BF16>) stream 4 KB of operands per 4 MFMAs per wave: with the bf16 MFMA 16x
// faster than the fp32 one that is 128 B/cycle/CU of L1/L2 traffic, and the kernel sits at ~20 % of the 2.5 PFLOP/s roof.  Here a
// workgroup owns a 128-row x 128- or 256-column tile; its operands are brought into LDS ONCE per 32-channel block and shared by
// the 4 / 8 waves (2 x 2 or 2 x 4, each 64 x 64 = 2 x 2 MFMA blocks of v_mfma_f32_32x32x16_bf16):
//
//   A (weights)      packed on the device at attach time as [m-tile][block][tap][k16-step][k-group][128 rows] x 16 B: one block is a
//                    contiguous 8 KB x taps slab, copied linearly global -> LDS; a lane's fragment (row i, 8 consecutive k) is one
//                    conflict-free ds_read_b128.
//   B (activations)  C8-blocked bf16 (16-byte group = 8 consecutive channels of one column = a lane's B fragment): per block the
//                    4 channel-group rows x (tile + 16) columns window is staged once and the conv's THREE TAPS read it at shifted
//                    columns -- the dilated-conv window in LDS that north_star asks for; a third of the activation traffic.
//   staging          LDS-DMA (global_load_lds_dwordx4: both LDS images are linear in the order the lanes address them), 2 stages
//                    x 33 KB and 2 workgroups per CU (128 columns) or 3 stages x 41 KB and 1 workgroup per CU (256 columns: 37 %
//                    fewer staged bytes per MFMA, a block's DMA runs two blocks ahead); one barrier per block; the DMA is counted by
//                    hand (s_waitcnt vmcnt(N)) and issued a piece at a time between the MFMA steps.
//
// Measured steps at batch 16 (conv + gate, us per launch, hipExt events): register-direct 75.9 -> LDS tile through staging registers
// 63.5 (register ARRAYS filled by loads went through scratch memory until they became named scalars: 185) -> loads-first epilogues
// 59.2 -> LDS-DMA 55.0 -> 256-column tile, 3 stages, pieces spread over the MFMA steps 51.9 -> asm DMA (clean lgkmcnt counting) 51.0
// = 34 % of the nominal 2.5 PFLOP/s roof; per workgroup 43 k shader cycles of which the MFMAs need 24.6 k (the chip runs this
// kernel at ~1.7 GHz).  What is left: the epilogue's fp32 conditioner reads (all CUs reach it together: an HBM burst), barrier skew.
//
// Paired rows (dilated conv + gate): rows 0..63 of a wave pair = gate rows of 32 channels (wave row wr) + the matching filter rows, so
// sigmoid(g) * tanh(f) is formed in registers.  Epilogues write the next GEMM's operand directly in the blocked 16-bit layout
// (bf_store_quad).  Split mode at batch 16: conv + gate 123.0 us (fp32 kernels: 333.6), out-projection 75.4 (144.9).
#pragma once
#include "convgemm.hip.h"
#include <cstdlib>
#include <type_traits>

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


// ---- epilogues: called once per (row block pair | row block, column block) with the wave's accumulators
struct BfEpiGate {     // wavenet.py:112-115; Z out as blocked bf16 (the out-projection's operand)
  static constexpr int kTaps = 3;
  static constexpr bool kPaired = true;
  const float* P; long p_bs; int ldp;      // conditioner slab (+ biases), fp32 [B][2C][ldp]
  void* Zb; long zb_bs; int ldz;           // blocked 16-bit out (bf_store_quad); zb_bs in elements
  int C;
  float acc_scale = 1.f, out_scale = 1.f;  // fp16-split mode: 2^-(operand scales) on the accumulators, 2^k on the stored operand
};
struct BfEpiResSkip {  // wavenet.py:117-120 + the skip sum of :228; Y = x + step out as blocked bf16 (the next conv's operand)
  static constexpr int kTaps = 1;
  static constexpr bool kPaired = false;
  float* X; float* SK; long bs; int ld;    // fp32 residual stream / skip sum [B][C][ld]
  const float* bias;                       // [2C]
  const float* sb; int sb_ld, sb_bs;       // next layer's diffusion projection
  void* Yb; long yb_bs;                    // blocked 16-bit out (null on the last layer)
  int C, skip_mode;
  float inv_div, r_inv_div;
  float acc_scale = 1.f, out_scale = 1.f;
  const float* keep = nullptr; long keep_bs = 0;   // exact-mask mode (EpiResSkip16S): the blocked output is 0 where keep[b][t] == 0
};

// ---- the blocked 16-bit operand layouts (16-byte group = 8 consecutive channels of one column = a lane's B fragment)
//   bf16 mode:        element (c, t) at ((c >> 3) * ld + t) * 8 + (c & 7)
//   fp16-split mode:  per 16-channel block four group rows [hi g0][hi g1][lo g0][lo g1]: hi at (((c >> 4) * 4 + ((c >> 3) & 1)) * ld + t) * 8
//                     + (c & 7), lo two rows further: value * 2^k = hi + lo with hi = fp16(value * 2^k), lo = fp16(value * 2^k - hi)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int F16S>
__device__ __forceinline__ void bf_store_quad(void* base, long item_off, int ld, int c0, int t, const float (&v)[4], float scale) {
  if constexpr (F16S) {
    f16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float sv = v[k] * scale;
      hi[k] = (_Float16)sv;
      lo[k] = (_Float16)(sv - (float)hi[k]);
    }
    _Float16* p = static_cast<_Float16*>(base) + item_off + ((long)((c0 >> 4) * 4 + ((c0 >> 3) & 1)) * ld + t) * 8 + (c0 & 7);
    *reinterpret_cast<f16x4*>(p) = hi;
    *reinterpret_cast<f16x4*>(p + (long)2 * ld * 8) = lo;
  } else {
    bf16x4 z;
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = (__bf16)v[k];
    *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(base) + item_off + ((long)(c0 >> 3) * ld + t) * 8 + (c0 & 7)) = z;
  }
}


typedef __attribute__((address_space(3))) void* bf_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* bf_glb_ptr_t;

// WN = waves along the columns (2: 128 x 128 tile, 4 waves, 2 LDS stages, 2 workgroups per CU; 4: 128 x 256 tile, 8 waves, 3 LDS
// stages, 1 workgroup per CU).  The conv's weight slab (3 taps) is 2.7x its activation window, so the tile grows along the columns:
// bytes staged per MFMA drop by 37 % and the third stage lets a block's DMA run two blocks ahead.
template <int WN> struct BfGeom {
  static constexpr int kWaves = 2 * WN, kThreads = 64 * kWaves, kCols = 64 * WN, kWin = kCols + 16, kStages = WN == 4 ? 3 : 2;
  static_assert((4 * kWin) % 64 == 0, "whole 1-KiB pieces per 4-row window");
};

// F16S = 1: the fp16-split mode ("past the fp32 roof"): every operand is a pair of fp16 numbers hi + lo (22 mantissa bits), the
// product block is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (the dropped lo.lo term is 2^-22 of
// the product) -- fp32-class results at a third of the fp16 MFMA rate = 5.3x the fp32 MFMA rate.  Same kernel, same LDS image:
// a block is 16 channels and the k16-step dimension of the bf16 image becomes the {hi, lo} dimension.
template <class Epi, int WN, int F16S = 0, int DBG = 0>
__global__ __launch_bounds__(BfGeom<WN>::kThreads, WN == 4 ? 1 : 2) void bf16lds_kernel(BfArgs a, Epi epi) {
  using Ge = BfGeom<WN>;
  constexpr int TAPS = Epi::kTaps, NW = Ge::kWaves, WIN = Ge::kWin, NST = Ge::kStages;
  // SUB: 16-channel sub-blocks per stage.  The split mode's out-projection (one tap) takes two: 12 MFMAs per wave between barriers
  // did not cover a stage's DMA and barrier (K loop 41.7 k cycles against 24.6 k of MFMAs).
  constexpr int SUB = (F16S && TAPS == 1) ? 2 : 1;
  constexpr int A_G = SUB * TAPS * 2 * 2 * 128;     // 16-byte groups of A per stage
  constexpr int B_G = SUB * 4 * WIN;                // ... of B: 4 group rows per (sub-)block
  constexpr int A_LD = A_G / (64 * NW);             // A pieces (1 KiB) per wave per stage
  constexpr int B_P = B_G / 64, QB = B_P / NW, RB = B_P % NW;   // B pieces: QB per wave, one more on waves < RB
  constexpr int STAGE_G = A_G + B_G;
  static_assert(A_G % (64 * NW) == 0 && B_G % 64 == 0, "whole pieces");
  __shared__ uint4 lds[NST * STAGE_G];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
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
  const int t0 = (nt - item * a.tiles_per_item) * Ge::kCols;

  const uint4* Ag = a.Wp + (size_t)mt * a.n_blk * A_G;
  const uint4* Bg = a.Xb + item * a.x_bs + (t0 - 8);

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][nb][r] = 0.f;

  // ---- staging: global -> LDS directly (global_load_lds_dwordx4: one wave-instruction lands 1 KiB at a wave-uniform LDS base +
  // lane * 16, which is exactly the linear image both operands have here).  No staging registers and no ds_write pass: through
  // registers the ds_write_b128s alone were 10.8 k of the K loop's 42.6 k cycles (13 cycles of the CU's VGPR -> LDS path each).
  size_t bo[QB + 1];
#pragma unroll
  for (int k = 0; k <= QB; ++k) {
    const int e = min((k * NW + wave) * 64 + lane, B_G - 1);
    const int row = e / WIN, col = e - row * WIN;
    bo[k] = (size_t)row * a.ld + col;
  }
  const size_t b_blk = (size_t)SUB * 4 * a.ld;
  constexpr int NP = A_LD + QB + 1;                 // DMA pieces per wave per stage (the last one on waves < RB only)
  // (inline asm, not __builtin_amdgcn_global_load_lds: with the builtin in flight hipcc stops counting its LDS reads and waits
  // lgkmcnt(0) in front of every other MFMA step; the asm statement is invisible to its counters -- the DMA is counted by hand below)
  const unsigned lds0 = (unsigned)(size_t)(bf_lds_ptr_t)lds + (unsigned)wave * 1024u;
  auto glds16 = [&](const uint4* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  auto piece = [&](int q, int bk, int st) {
    const unsigned l = lds0 + (unsigned)st * (STAGE_G * 16);
    const uint4* pb = Bg + (size_t)bk * b_blk;
    if (q < A_LD) glds16(Ag + (size_t)bk * A_G + tid + q * NW * 64, l + q * NW * 1024);
    else if (q < A_LD + QB) glds16(pb + bo[q - A_LD], l + (A_G + (q - A_LD) * NW * 64) * 16);
    else if (wave < RB) glds16(pb + bo[QB], l + (A_G + QB * NW * 64) * 16);
  };
  // The DMA is waited for by hand: __syncthreads() would drain it (vmcnt(0)) at every barrier; with three stages the newest block
  // stays in flight across the barrier (its pieces: A_LD + QB, + 1 on waves < RB).
  auto wait_landed = [&](bool newest_in_flight) {
    if (NST == 3 && newest_in_flight) {
      if (wave < RB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + QB + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + QB) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // One block's MFMAs, with the DMA of block bk2 (into stage st2) issued a piece or two per MFMA step: an LDS-DMA piece costs its
  // wave 60-180 issue cycles, which hide behind the previous step's MFMAs (asynchronous: 128 pipe cycles per step) but not when
  // all of a block's pieces are issued back to back in front of them (measured: +10 k cycles on a 29.5 k K loop).
  // The fragment reads are software-pipelined by hand too (step j+1's four ds_read_b128 before step j's four MFMAs): left to itself
  // hipcc re-uses ONE fragment register set, so every step waits out the LDS latency.
  auto compute = [&](int st, int bk2, int st2) {
    const uint4* la = lds + st * STAGE_G + wr * 64 + i + g * 128;
    const uint4* lb = lds + st * STAGE_G + A_G + g * WIN + wc * 64 + i;
    constexpr int NS = F16S ? TAPS * SUB : TAPS * 2;          // MFMA steps per stage: a tap of a sub-block (both halves of the split) | a k16-step of a tap
    constexpr int SPREAD = NST == 3 ? NS : (NS + 1) / 2;      // two stages: the pieces must land before this block's barrier
    constexpr int NF = F16S ? 2 : 1;                          // fragment pairs per step: {hi, lo} | one
    uint4 fa[2][NF][2], fb[2][NF][2];
    auto frag = [&](int j, int set) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int u = (F16S && SUB == 2) ? j : 0;             // sub-block (one tap then)
        const int tap = F16S ? (SUB == 2 ? 0 : j) : j >> 1, s2 = F16S ? f : j & 1;
        const int shift = TAPS == 3 ? 8 + (tap - 1) * a.dil : 8;
#pragma unroll
        for (int x = 0; x < 2; ++x) fa[set][f][x] = la[((u * TAPS + tap) * 2 + s2) * 256 + x * 32];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) fb[set][f][nb] = lb[(u * 4 + 2 * s2) * WIN + shift + nb * 32];
      }
    };
    if (!(DBG & 4)) frag(0, 0);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (!(DBG & 4) && j + 1 < NS) frag(j + 1, (j + 1) & 1);
      if (!(DBG & 1) && j < SPREAD) {
#pragma unroll
        for (int q = j * NP / SPREAD; q < (j + 1) * NP / SPREAD; ++q) piece(q, bk2, st2);
      }
      __builtin_amdgcn_sched_barrier(0);            // (pins the order: the scheduler otherwise undoes the pipelining)
      if (!(DBG & 4)) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            if constexpr (F16S) {                   // cross terms first, then the leading one
              const f16x8 ah = __builtin_bit_cast(f16x8, fa[j & 1][0][x]), al = __builtin_bit_cast(f16x8, fa[j & 1][1][x]);
              const f16x8 bh = __builtin_bit_cast(f16x8, fb[j & 1][0][nb]), bl = __builtin_bit_cast(f16x8, fb[j & 1][1][nb]);
              acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[x][nb], 0, 0, 0);
              acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[x][nb], 0, 0, 0);
              acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[x][nb], 0, 0, 0);
            } else {
              acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[j & 1][0][x]),
                                                                   __builtin_bit_cast(bf16x8, fb[j & 1][0][nb]), acc[x][nb], 0, 0, 0);
            }
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- K loop over the 32-channel blocks; the stage indices are compile-time constants (unrolled NST-fold) so that the DMA into
  // one stage and the ds_reads of another are provably disjoint.  Past the last block the DMA re-loads the last block into a stage
  // nobody reads again: every iteration issues the same number of pieces, so the counted wait holds to the end.
  const int last = a.n_blk - 1;
  FDX_STAMP(0);
  FDX_STAMP_RT0();
  if (!(DBG & 1)) {
#pragma unroll
    for (int q = 0; q < NP; ++q) piece(q, 0, 0);
    if (NST == 3) {
#pragma unroll
      for (int q = 0; q < NP; ++q) piece(q, min(1, last), 1);
    }
  }
  wait_landed(true);
  FDX_STAMP(1);
  auto body = [&](auto S_, int blk) {
    constexpr int S = decltype(S_)::value;
    compute(S, min(blk + NST - 1, last), (S + NST - 1) % NST);
    wait_landed(true);
  };
  for (int blk = 0; blk < a.n_blk; blk += NST) {
    body(std::integral_constant<int, 0>{}, blk);
    if (blk + 1 < a.n_blk) body(std::integral_constant<int, 1>{}, blk + 1);
#ifdef FDX_KTRACE
    if (blk == 0) FDX_STAMP(3);
#endif
    if (NST == 3 && blk + 2 < a.n_blk) body(std::integral_constant<int, NST - 1>{}, blk + 2);
#ifdef FDX_KTRACE
    if (blk == 0) FDX_STAMP(4);
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing re-loads must have landed before this workgroup's LDS is released
  FDX_STAMP(2);

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
        float z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = q * 4 + k;
          z[k] = EpiGate::gate1(acc[0][nb][r] * epi.acc_scale + pg[r], acc[1][nb][r] * epi.acc_scale + pf[r]);
        }
        bf_store_quad<F16S>(epi.Zb, item * epi.zb_bs, epi.ldz, ch0 + 8 * q + 4 * half, t, z, epi.out_scale);   // (the quad's first channel: acc_row(4q, half))
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
        const float* kp = (use_sb && epi.keep) ? epi.keep + item * epi.keep_bs + t : epi.bias;
        const float kraw = *kp;
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
          float y[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = q * 4 + k;
            float v = acc[x][nb][r] * epi.acc_scale + bi[r];
            if (res) {
              v = div_const(old[r] + v, 1.41421356237309504880f, 0.70710678118654752440f);
              y[k] = (epi.Yb && (!epi.keep || kraw != 0.f)) ? v + sbv[r] : 0.f;
            } else {
              if (rd) v = old[r] + v;
              if (epi.skip_mode >= 2) v = div_const(v, epi.inv_div, epi.r_inv_div);
            }
            RW[o0 + (long)acc_row(r, half) * epi.ld] = v;
          }
          if (res && epi.Yb) bf_store_quad<F16S>(epi.Yb, item * epi.yb_bs, epi.ld, row0 + 8 * q + 4 * half, t, y, epi.out_scale);
        }
      }
    }
  }
  FDX_STAMP_END();
}

template <class Epi, int WN, int F16S>
inline hipError_t launch_bf16lds_wn(BfArgs a, int B, int T, const Epi& epi, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
  using Ge = BfGeom<WN>;
  a.tiles_per_item = (T + Ge::kCols - 1) / Ge::kCols;
  a.n_tiles_n = B * a.tiles_per_item;
  const int grid = a.n_tiles_n * a.n_mtiles;
  if (grid <= 0) return hipSuccess;
#ifdef FDX_KTRACE
  a.trace = nullptr;
  if (g_trace.buf && g_trace.n < g_trace.max_launches && grid <= g_trace.blocks_cap)
    a.trace = g_trace.buf + (size_t)(g_trace.n++) * g_trace.blocks_cap * 32;
  static const int dbg = [] { const char* e = getenv("FDX_BF16_DBG"); return e ? atoi(e) : 0; }();
#define FDX_BF_DBG(D) if (dbg == D) { hipLaunchKernelGGL((bf16lds_kernel<Epi, WN, F16S, D>), dim3(grid), dim3(Ge::kThreads), 0, s, a, epi); return hipGetLastError(); }
  FDX_BF_DBG(1) FDX_BF_DBG(4) FDX_BF_DBG(5)
#undef FDX_BF_DBG
#endif
  if (ev0) hipExtLaunchKernelGGL((bf16lds_kernel<Epi, WN, F16S>), dim3(grid), dim3(Ge::kThreads), 0, s, ev0, ev1, 0, a, epi);
  else hipLaunchKernelGGL((bf16lds_kernel<Epi, WN, F16S>), dim3(grid), dim3(Ge::kThreads), 0, s, a, epi);
  return hipGetLastError();
}

// FDX_BF16_WN = 2 | 4 forces a tile width (A/B).  The wide tile runs one workgroup per CU in lock-step rounds of 256: it is chosen
// when its rounds are at least 80 % full (measured, 50 steps: batch 8 = one full round 59 vs 63 ms; batch 12 = 1.5 rounds 94 vs 92;
// batch 16 = two full rounds 286 vs 300 ms per 100 steps).
inline int bf16lds_pick_wn(int B, int T, int rows) {
  static const int forced = [] { const char* e = getenv("FDX_BF16_WN"); return e ? atoi(e) : 0; }();
  if (forced == 2 || forced == 4) return forced;
  const long wide = (long)B * ((T + 255) / 256) * (rows / 128), rounds = (wide + 255) / 256;
  return wide * 5 >= rounds * 256 * 4 ? 4 : 2;
}

// x_bs: 16-byte groups between items of Xb.  F16S = 0: bf16 operands, blocks of 32 channels; 1: fp16 {hi, lo} operands, blocks of 16.
template <int F16S = 0, class Epi>
inline hipError_t launch_bf16lds(const uint4* Wp, const uint4* Xb, long x_bs, int ld, int C, int dil, int B, int T, int rows, const Epi& epi,
                                 hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
  BfArgs a;
  a.Wp = Wp; a.Xb = Xb; a.x_bs = x_bs; a.ld = ld; a.dil = dil; a.T = T;
  a.n_blk = C / ((F16S && Epi::kTaps == 3) ? 16 : 32);     // channels per LDS stage (the split mode's one-tap GEMM stages two 16-channel sub-blocks)
  a.n_mtiles = rows / 128;
  a.tiles_per_item = a.n_tiles_n = 0;
  return bf16lds_pick_wn(B, T, rows) == 4 ? launch_bf16lds_wn<Epi, 4, F16S>(a, B, T, epi, s, ev0, ev1)
                                          : launch_bf16lds_wn<Epi, 2, F16S>(a, B, T, epi, s, ev0, ev1);
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
