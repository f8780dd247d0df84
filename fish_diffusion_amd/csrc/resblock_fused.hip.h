// resblock_fused.hip.h -- one launch per ResBlock1 for the vocoders' SMALL-CHANNEL stages (C = 16 / 32 at 2-4 x 10^5 samples).
//
// Reference: fish_diffusion/modules/vocoders/nsf_hifigan/models.py:44-110 (ResBlock1: three pairs  xt = c1_j(lrelu(x)); xt = c2_j(lrelu(xt));
// x = xt + x, c1 dilated by d_j, c2 by 1) and :426-432 (the MRF mean over the three kernel sizes); RefineGAN's ResBlock
// (refinegan/generator.py:63-75: both convs of a pair dilated by d_j, slope 0.2).
//
// Why: run conv by conv, these stages are not MFMA-bound but traffic-bound -- each of the 18 convs of a stage reads 28 MB, writes 28 MB and
// the residual adds another read (2.9-3.5 TB/s measured at C = 16, profiles/r03_vocoder_launch_sequence.txt: 27 % / 47 % of the fp32 MFMA
// roof).  Here a workgroup owns N output columns of ALL C channels, stages the window x[C][N + 2H] (H = the six convs' receptive field:
// 12 / 36 / 60 columns per side for k = 3 / 7 / 11) ONCE into LDS and runs the six convs out of LDS -- the intermediate of each pair in a
// second LDS buffer, the running x updated in place -- so a ResBlock reads its input once and writes its output once.  The halo columns are
// recomputed per tile (N = 1024 at C = 16: +6 % MFMA work on average; N = 448 at C = 32: +13 %), a trade the roofs make easy.
//
//   v_mfma_f32_16x16x4_f32:  A = weights [16 rows][4 channels] of one tap (lane l: row l & 15, channel l >> 4) -- a conv's whole weight set
//   lives in registers (12 ... 176 per lane), loaded once per conv per tile; B = activations [4 channels][16 columns] read from LDS with
//   ds_read2_b32 (lane l: channel l >> 4, column l & 15; row pitch = 16 mod 32 banks -> conflict-free); D = exactly C rows: no row padding.
//   A wave owns 32-column units (2 x C/16 independent accumulators) dealt round-robin; one barrier per conv.
//
// Zero padding of the reference's convs applies to EVERY conv's input separately: every intermediate is written as 0 outside [0, L).
// Summation order differs from the per-conv kernels (tap-major over 4-channel groups instead of 8-channel blocks): fp32-rounding-level
// differences, inside the 1e-4 waveform bar with a margin of 30 (tests/test_gpu_round4.py compares the two paths directly).
#pragma once
#include <type_traits>

#include "common.hip.h"

#ifndef FDX_RB_EXP
#define FDX_RB_EXP 0     // timing experiments of tools/ubench/rbfused.hip (results wrong): 1 no leaky-relu, 2 no epilogue, 4 no LDS reads in the K loop
#endif

namespace fdx {

constexpr int kRbPairs = 3;   // (c1, c2) pairs per ResBlock1: the shipped configs' [1, 3, 5]

struct RbFusedArgs {
  const float* X; long x_bs; int ldx;     // input rows [B][C][ldx], pointer at column 0
  float* out; long o_bs; int ldo;         // output rows
  const float* W;                         // this ResBlock's fused weights: 6 convs x [tap][rb][q][lane] float4 (rb_fused_pack)
  const float* bias;                      // 6 x C
  int L, N, tiles_per_item;               // valid columns, owned columns per tile (multiple of 32)
  int d1[kRbPairs], d2[kRbPairs];         // dilations of c1_j / c2_j
  float slope;
  int mode; float div;                    // output: 0: out = v   1: out += v   2: out = (out + v) / div   (EpiResblock's modes)
  unsigned long long* trace;              // FDX_RB_TRACE builds (tools/ubench/rbfused.hip): 9 shader-clock stamps per wave, else unused
};

template <int C> struct RbGeom {
  static constexpr int P = C == 16 ? 1168 : 592;     // LDS row pitch in floats: 16 mod 32, two buffers of C x P fit 160 KB
};
inline int rb_halo(int ks, const int* d1, const int* d2) {
  int h = 0;
  for (int j = 0; j < kRbPairs; ++j) h += (ks - 1) / 2 * (d1[j] + d2[j]);
  return h;
}
// owned columns per tile: the widest window the LDS rows hold, then shrunk so that the tiles fill whole rounds of `n_cu` workgroups
// (the chip runs ceil(tiles / n_cu) rounds of one workgroup per CU; each costs ~ N + 2H columns of work)
inline int rb_pick_n(int C, int H4, long L, int B, int n_cu = 256) {
  const int P = C == 16 ? RbGeom<16>::P : RbGeom<32>::P;
  const int n_max = (P - 32 - 2 * H4) / 32 * 32;
  if (n_max < 64) return 0;
  int best = n_max;
  double best_cost = 1e300;
  for (int n = n_max; n >= 64 && n >= n_max / 2; n -= 32) {
    const long tiles = (long)B * ((L + n - 1) / n);
    const double cost = (double)((tiles + n_cu - 1) / n_cu) * (n + 2.0 * H4);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = n; }
  }
  return best;
}

// host: Conv1d weight w[C][C][KS] -> [tap][rb][q][lane][e]: row rb*16 + (lane & 15), channel (4*q + e)*4 + (lane >> 4)
inline void rb_fused_pack(float* dst, const float* w, int C, int KS) {
  const int RB = C / 16, Q = C / 16;
  for (int tap = 0; tap < KS; ++tap)
    for (int rb = 0; rb < RB; ++rb)
      for (int q = 0; q < Q; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int row = rb * 16 + (lane & 15), ch = (4 * q + e) * 4 + (lane >> 4);
            dst[((((size_t)tap * RB + rb) * Q + q) * 64 + lane) * 4 + e] = w[((size_t)row * C + ch) * KS + tap];
          }
}
inline size_t rb_fused_floats(int C, int KS) { return (size_t)KS * C * C; }

#ifndef FDX_RB_WAVES
#define FDX_RB_WAVES 8
#endif
constexpr int kRbWaves = FDX_RB_WAVES;   // waves per workgroup: two per SIMD -- with one, nothing hides a wave's LDS reads / epilogue / address arithmetic (measured additive)

template <int C, int KS>
__global__ __launch_bounds__(kRbWaves * 64) void k_resblock1_fused(RbFusedArgs a) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int RB = C / 16, Q = C / 16, CG = C / 4, P = RbGeom<C>::P, KH = (KS - 1) / 2, NA4 = KS * Q;
  // a wave holds the weights of ONE 16-row block (C = 32: waves 0, 2 the rows 0-15, waves 1, 3 the rows 16-31; a unit is then computed by a pair
  // of waves, each reading the same B values) -- both row blocks of k = 11 would be 176 registers per lane and spill
  constexpr int UW = kRbWaves / RB, NT = kRbWaves * 64;   // waves that share a row block = unit stride; threads
  constexpr bool PREF = NA4 * 4 <= 64;           // small weight sets: the NEXT conv's weights are requested while this conv computes
  __shared__ float sx[C * P + 128];              // running x (raw)
  __shared__ float st[C * P + 128];              // the pair's intermediate, stored leaky-relu'd (only c2 reads it)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lk = lane >> 4;
  const int rb = RB == 1 ? 0 : (wave & (RB - 1)), uw = wave / RB;      // this wave's row block; its first unit
  const int item = blockIdx.x / a.tiles_per_item, t0 = (blockIdx.x - item * a.tiles_per_item) * a.N;
  // (scalars, not the argument struct's arrays: indexing those -- even with unrolled constants -- made hipcc keep the struct in scratch memory)
  const int d1_[kRbPairs] = {a.d1[0], a.d1[1], a.d1[2]}, d2_[kRbPairs] = {a.d2[0], a.d2[1], a.d2[2]};
  const int H = KH * (d1_[0] + d1_[1] + d1_[2] + d2_[0] + d2_[1] + d2_[2]);
  const int H4 = (H + 3) & ~3, N = a.N, W = N + 2 * H4, g0 = t0 - H4;     // local column u <-> global column g0 + u
  const float slope = a.slope;
#ifdef FDX_RB_TRACE
  unsigned long long stamps[9];
  int n_st = 0;
#define RB_STAMP() stamps[n_st++] = __builtin_amdgcn_s_memtime()
#else
#define RB_STAMP() do { } while (0)
#endif
  RB_STAMP();

  // ---- weights of a conv: NA4 float4 per lane, fragment order
  auto load_w = [&](f4 (&A)[NA4], int conv) __attribute__((always_inline)) {
    const f4* src = reinterpret_cast<const f4*>(a.W) + (size_t)conv * (KS * RB * Q * 64) + lane;
#pragma unroll
    for (int i = 0; i < NA4; ++i) A[i] = src[(size_t)(((i / Q) * RB + rb) * Q + (i % Q)) * 64];   // i = tap * Q + q
  };
  f4 A0[NA4], A1[PREF ? NA4 : 1];
  load_w(A0, 0);

  // ---- the window: x[C][W] -> LDS, zero outside [0, L).  Batches of independent, unconditional 16-byte loads (a group outside the signal reads
  // column 0 of its row instead and is zeroed by a select): with the bounds test around the load hipcc waited for every load before issuing
  // the next -- 18 exposed HBM round trips per tile.  L and the window origin are multiples of 4: a group is wholly inside or wholly outside.
  {
    const int w4 = W >> 2, total = C * w4;
    const float* Xb = a.X + item * a.x_bs;
    constexpr int BATCH = 3;
    for (int base = tid; base < total; base += NT * BATCH) {
      f4 v[BATCH];
      int dst[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int idx = min(base + NT * k, total - 1);
        const int c = idx / w4, g = idx - c * w4, gt = g0 + 4 * g;
        const bool in = gt >= 0 && gt < a.L;
        v[k] = *reinterpret_cast<const f4*>(Xb + (long)c * a.ldx + (in ? gt : 0));
        if (!in) v[k] = f4{0.f, 0.f, 0.f, 0.f};
        dst[k] = c * P + 4 * g;
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k)
        if (base + NT * k < total) *reinterpret_cast<f4*>(sx + dst[k]) = v[k];
    }
  }
  __syncthreads();
  RB_STAMP();

  // ---- one conv over local columns [lo, hi): src -> (epilogue)
  //   IS_C2 = false: st = lrelu(conv(lrelu(sx)) + b)            IS_C2 = true, !LAST: sx = conv(st) + b + sx
  //   LAST: out (global) = conv(st) + b + sx under a.mode, owned columns only
  // (always_inline: called three times per instantiation; out of line, the weight array it takes by reference would live in scratch memory)
  auto conv_pass = [&](auto IS_C2_, auto LAST_, const f4 (&A)[NA4], int conv, int dil, int lo, int hi) __attribute__((always_inline)) {
    constexpr bool IS_C2 = decltype(IS_C2_)::value, LAST = decltype(LAST_)::value;
    const float* src = IS_C2 ? st : sx;
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = a.bias[conv * C + rb * 16 + 4 * lk + r];
    const int n_units = (hi - lo + 31) >> 5;
    for (int unit = uw; unit < n_units; unit += UW) {
      const int u0 = lo + 32 * unit;
      f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
      float old[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      if constexpr (LAST) {            // the MRF sum's running value: requested here, used behind the K loop (unconditional loads, clamped column)
        if (a.mode != 0) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int gt = min(max(g0 + u0 + 16 * m + lj, 0), a.L - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) old[m][r] = a.out[item * a.o_bs + (long)(rb * 16 + 4 * lk + r) * a.ldo + gt];
          }
        }
      }
      // K loop, tap-major: the B values of tap + 1 (CG x 2 LDS reads) are requested before the MFMAs of tap issue -- left to itself hipcc put
      // every ds_read directly in front of its MFMA behind an lgkmcnt(0), i.e. one exposed LDS round trip per MFMA pair
      const float* bp = src + lk * P + u0 + lj - KH * dil;
      float bq[2][CG][2];
      auto rd = [&](int buf, const float* p) __attribute__((always_inline)) {
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
          if constexpr ((FDX_RB_EXP & 4) != 0) { bq[buf][cg][0] = slope + (float)cg; bq[buf][cg][1] = slope - (float)cg; }
          else { bq[buf][cg][0] = p[cg * 4 * P]; bq[buf][cg][1] = p[cg * 4 * P + 16]; }
        }
      };
      rd(0, bp);
#pragma unroll
      for (int tap = 0; tap < KS; ++tap) {
        if (tap + 1 < KS) rd((tap + 1) & 1, bp + (tap + 1) * dil);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
          float b0 = bq[tap & 1][cg][0], b1 = bq[tap & 1][cg][1];
          if constexpr (!IS_C2 && (FDX_RB_EXP & 1) == 0) {      // leaky-relu on the operand: max(x, slope x) == the per-conv kernels' select for slope < 1
            // max(x, slope x) as v_mul + v_max: __builtin_fmaxf (and fmed3 with +inf, which folds to it) adds a canonicalising v_max per operand, and
            // every VALU op here costs MFMA time.  Inline asm, so the VALU-write -> MFMA-read hazard is ours to cover: the hazard recogniser does
            // not see inside an asm statement (without the s_nop the MFMA read the operand too early: wrong results, caught by the ubench's check).
            const float t0_ = b0 * slope, t1_ = b1 * slope;
            float r0_, r1_;
            asm("v_max_f32 %0, %2, %4\n\tv_max_f32 %1, %3, %5\n\ts_nop 1" : "=&v"(r0_), "=&v"(r1_) : "v"(b0), "v"(b1), "v"(t0_), "v"(t1_));
            b0 = r0_; b1 = r1_;
          }
          const float av = A[tap * Q + (cg >> 2)][cg & 3];
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue: lane holds rows rb*16 + 4*lk + r of column u0 + 16*m + lj
      if constexpr ((FDX_RB_EXP & 2) != 0) { if (acc[0][0] + acc[1][1] == 1.2345f) st[lane] = acc[0][2]; continue; }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int u = u0 + 16 * m + lj, gt = g0 + u;
        const bool inside = gt >= 0 && gt < a.L;
        if (u >= hi) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = rb * 16 + 4 * lk + r;
            float v = acc[m][r] + bias[r];
            if constexpr (!IS_C2) {
              v = v > 0.f ? v : v * slope;
              st[row * P + u] = inside ? v : 0.f;
            } else if constexpr (!LAST) {
              v = v + sx[row * P + u];
              sx[row * P + u] = inside ? v : 0.f;
            } else {
              if (inside) {
                v = v + sx[row * P + u];
                float* o = a.out + item * a.o_bs + (long)row * a.ldo + gt;
                if (a.mode == 1) v = old[m][r] + v;
                else if (a.mode == 2) v = (old[m][r] + v) / a.div;
                __builtin_nontemporal_store(v, o);
              }
            }
          }
      }
    }
  };

  int rem = H;
  auto run_pair = [&](auto J_) __attribute__((always_inline)) {
    constexpr int j = decltype(J_)::value;
    // c1_j
    rem -= KH * d1_[j];
    if constexpr (PREF) load_w(A1, 2 * j + 1);
    conv_pass(std::false_type{}, std::false_type{}, A0, 2 * j, d1_[j], H4 - rem, H4 + N + rem);
    if constexpr (!PREF) load_w(A0, 2 * j + 1);
    __syncthreads();
    RB_STAMP();
    // c2_j
    rem -= KH * d2_[j];
    if constexpr (PREF) {
      if constexpr (j + 1 < kRbPairs) load_w(A0, 2 * j + 2);
      conv_pass(std::true_type{}, std::integral_constant<bool, j + 1 == kRbPairs>{}, A1, 2 * j + 1, d2_[j], H4 - rem, H4 + N + rem);
    } else {
      conv_pass(std::true_type{}, std::integral_constant<bool, j + 1 == kRbPairs>{}, A0, 2 * j + 1, d2_[j], H4 - rem, H4 + N + rem);
      if constexpr (j + 1 < kRbPairs) load_w(A0, 2 * j + 2);
    }
    if constexpr (j + 1 < kRbPairs) __syncthreads();
    RB_STAMP();
  };
  run_pair(std::integral_constant<int, 0>{});
  run_pair(std::integral_constant<int, 1>{});
  run_pair(std::integral_constant<int, 2>{});
#ifdef FDX_RB_TRACE
  if (a.trace && lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a.trace[((size_t)blockIdx.x * kRbWaves + wave) * 8 + i] = stamps[i];
  }
#endif
#undef RB_STAMP
}

// true if a fused instantiation exists for (C, KS) and the geometry fits
// Instantiated for C = 16 and 32; TAKEN for C = 16 only (rb_fused_wins).  Measured on MI355X (tools/ubench/rbfused.hip, profiles/r04_resblock_fused_ubench.txt),
// one ResBlock1 at batch 1, us, fused | conv by conv:  C = 16 (L = 440 832): k = 3 / 7 / 11: 75 / 137 / 198 = 410 | 660;  C = 32 (L = 220 416): 133 / 254 / 418 = 805 | 765.
// At C = 32 two LDS buffers leave N = 288 ... 448 owned columns per tile: the recomputed halo (+21 ... +40 % MFMA work) and the one-row-block-per-wave
// split (every B value feeds ONE 32-cycle MFMA: the leaky-relu VALU ops and the LDS reads cost as much per MFMA as at C = 16) eat the traffic saved.
inline bool rb_fused_supported(int C, int KS) { return (C == 16 || C == 32) && (KS == 3 || KS == 7 || KS == 11); }
inline bool rb_fused_wins(int C) { return C == 16; }

inline hipError_t launch_resblock1_fused(int C, int KS, RbFusedArgs a, int B, hipStream_t s) {
  const int H = rb_halo(KS, a.d1, a.d2), H4 = (H + 3) & ~3;
  a.N = rb_pick_n(C, H4, a.L, B);
  if (a.N <= 0) return hipErrorInvalidValue;
  a.tiles_per_item = (a.L + a.N - 1) / a.N;
  const dim3 grid((unsigned)(B * a.tiles_per_item)), blk(kRbWaves * 64);
#define FDX_RB(C_, K_) if (C == C_ && KS == K_) { hipLaunchKernelGGL((k_resblock1_fused<C_, K_>), grid, blk, 0, s, a); return hipGetLastError(); }
  FDX_RB(16, 3) FDX_RB(16, 7) FDX_RB(16, 11) FDX_RB(32, 3) FDX_RB(32, 7) FDX_RB(32, 11)
#undef FDX_RB
  return hipErrorInvalidValue;
}

}  // namespace fdx
