// f16s64.hip.h -- the fp16-split mode (bf16lds.hip.h, F16S) for SMALL column counts: a 64-row x 64-column workgroup tile on
// v_mfma_f32_16x16x32_f16, so that batch 1 at 10 s (1024 rows x 861 columns) gives 16 x 14 = 224 workgroups instead of the 56 the
// 128 x 128 tile has there.  Same arithmetic contract (operands hi + lo in fp16 with power-of-two scales, product block =
// hi.lo + lo.hi + hi.hi, fp32 accumulate), same blocked {hi, lo} activation layout (bf_store_quad<1>), same LDS-DMA staging with a
// hand-counted three-stage pipeline; what differs is the fragment geometry:
//
//   MFMA 16x16x32:  A fragment = 16 rows x 32 k: lane l holds row (l & 15), k-group (l >> 4) (8 consecutive k = 16 bytes);
//                   B fragment = 32 k x 16 columns: lane l holds column (l & 15), k-group (l >> 4);
//                   C = 16 x 16: lane l holds column (l & 15), rows 4 * (l >> 4) .. + 3.
//   workgroup       4 waves as 2 (rows) x 2 (columns); a wave owns 32 rows x 32 columns = 2 x 2 MFMA blocks.  Paired rows (conv + gate):
//                   a wave's two row blocks are the gate rows and the filter rows of the same 16 channels.
//   stage           one 32-channel block: A [tap][hl][k-group 4][64 rows] x 16 B (weights packed in this order on the device:
//                   k_f16s64_from_arena), B = the 8 group rows x (64 + 16) columns of the blocked activations exactly as they lie in
//                   memory ([hi g0][hi g1][lo g0][lo g1] of the two 16-channel halves), 34 KB per stage, three stages.
//
// Status (round 3): on by default in the fp16-split mode for every launch below 200 wide tiles (wavenet.hip: f16s_small_min_tiles); held to the
// same broad parity subset as the wide tiles with this kernel forced for every geometry (42 tests: reference goldens incl. the full-size
// 1000-step DDPM fixtures, chained waveforms, exact-ragged batches).  Batch 1 x 10 s: 24.6 ms per 50 denoiser calls against 37.1 on the
// fp32 kernels; `python bench.py --storage fp16x3`: 179x real-time per GPU.  DESIGN.md section 5 has what bounds it (LDS-DMA, ~5 us fixed).
#pragma once
#include "bf16lds.hip.h"

namespace fdx {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kS64Waves = 4;
constexpr int kS64StagesConv = 3, kS64StagesOutp = 4;   // defaults (measured, 1 x 10 s, ms per 50 calls: 3/3 27.07, 3/4 26.09, 4/4 26.62, 3/6 26.93, 4/8 27.32)

template <int... I, class F>
__device__ __forceinline__ void s64_for_each_stage(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// NST: LDS stages (a stage is 34 KB for the conv, 18 KB for the one-tap out-projection).  The K loop is bound by LDS-DMA latency, not
// by its MFMAs (tools/ktrace.py: 1340 cycles per 32-channel block of the conv against 576 cycles of MFMA issue with three stages, i.e.
// one stage of look-ahead), so the stage count is how many blocks the DMA runs ahead: NST - 2 whole stages stay in flight across a barrier.
template <class Epi, int NST = 3>
__global__ __launch_bounds__(256, 1) void f16s64_kernel(BfArgs a, Epi epi) {
  constexpr int TAPS = Epi::kTaps, NW = kS64Waves;
  constexpr int PAD = TAPS == 3 ? 8 : 0;            // columns staged either side of the tile: the dilated taps reach 8; the one-tap GEMM none
  constexpr int WIN = 64 + 2 * PAD;                 // (out-projection without pads: 18 -> 16 KB per stage, 10.7 -> 9.3 us per launch with 4 stages)
  static_assert(NST >= 3, "at least three stages");
  constexpr int A_G = TAPS * 2 * 4 * 64;            // 16-byte groups of A per stage: [tap][hl][kg][64 rows]
  constexpr int B_G = 8 * WIN;                      // ... of B: 8 group rows of the 32-channel block
  constexpr int A_LD = A_G / (64 * NW);             // A pieces (1 KiB) per wave per stage
  constexpr int B_P = B_G / 64, QB = B_P / NW, RB = B_P % NW;
  constexpr int STAGE_G = A_G + B_G;
  static_assert(A_G % (64 * NW) == 0 && B_G % 64 == 0, "whole pieces");
  __shared__ uint4 lds[NST * STAGE_G];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int kg = lane >> 4, li = lane & 15;

  // ---- tile -> XCD map (as bf16lds_kernel)
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
  const int t0 = (nt - item * a.tiles_per_item) * 64;

  const uint4* Ag = a.Wp + (size_t)mt * a.n_blk * A_G;
  const uint4* Bg = a.Xb + item * a.x_bs + (t0 - PAD);

  f32x4_t acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[x][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- staging: LDS-DMA, one 1-KiB piece per wave-instruction (see bf16lds_kernel)
  size_t bo[QB + 1];
#pragma unroll
  for (int k = 0; k <= QB; ++k) {
    const int e = min((k * NW + wave) * 64 + lane, B_G - 1);
    const int row = e / WIN, col = e - row * WIN;
    bo[k] = (size_t)row * a.ld + col;
  }
  const size_t b_blk = (size_t)8 * a.ld;            // a 32-channel block = 8 group rows of the blocked activations
  constexpr int NP = A_LD + QB + 1;
  const unsigned lds0 = (unsigned)(size_t)(bf_lds_ptr_t)lds + (unsigned)wave * 1024u;
  auto glds16 = [&](const uint4* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  auto piece = [&](int q, int bk, int st) {
    const unsigned l = lds0 + (unsigned)st * (STAGE_G * 16);
    const uint4* pb = Bg + (size_t)bk * b_blk;
    if (q < A_LD) glds16(Ag + (size_t)bk * A_G + tid + q * NW * 64, l + q * NW * 1024);   // (an `nt` policy on this weight stream measured 5 % slower)
    else if (q < A_LD + QB) glds16(pb + bo[q - A_LD], l + (A_G + (q - A_LD) * NW * 64) * 16);
    else if (wave < RB) glds16(pb + bo[QB], l + (A_G + QB * NW * 64) * 16);
  };
  auto wait_landed = [&]() {                          // the newest NST - 2 stages stay in flight across the barrier
    static_assert((NST - 2) * (A_LD + QB + 1) <= 63, "vmcnt is a 6-bit counter");
    if (wave < RB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (A_LD + QB + 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (A_LD + QB)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- one stage's MFMAs: per tap 8 fragment reads (A: {hi, lo} x 2 row blocks, B: {hi, lo} x 2 column blocks) and 12 MFMAs
  // (cross terms first, then the leading one), the next tap's fragments read ahead; the next-but-one stage's DMA pieces are issued
  // between the taps.
  auto compute = [&](int st, int bk2, int st2) {
    const uint4* la = lds + st * STAGE_G + kg * 64 + wr * 32 + li;
    // B: k-group kg of half hl lives in group row (kg >> 1) * 4 + hl * 2 + (kg & 1) of the block
    const uint4* lb = lds + st * STAGE_G + A_G + ((kg >> 1) * 4 + (kg & 1)) * WIN + wc * 32 + li;
    uint4 fa[2][2][2], fb[2][2][2];                  // [set][hl][x | nb]
    auto frag = [&](int tap, int set) {
      const int shift = TAPS == 3 ? PAD + (tap - 1) * a.dil : 0;
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
        for (int x = 0; x < 2; ++x) fa[set][hl][x] = la[(tap * 2 + hl) * 256 + x * 16];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) fb[set][hl][nb] = lb[hl * 2 * WIN + shift + nb * 16];
      }
    };
    frag(0, 0);
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
      if (j + 1 < TAPS) frag(j + 1, (j + 1) & 1);
#pragma unroll
      for (int q = j * NP / TAPS; q < (j + 1) * NP / TAPS; ++q) piece(q, bk2, st2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const f16x8 ah = __builtin_bit_cast(f16x8, fa[j & 1][0][x]), al = __builtin_bit_cast(f16x8, fa[j & 1][1][x]);
          const f16x8 bh = __builtin_bit_cast(f16x8, fb[j & 1][0][nb]), bl = __builtin_bit_cast(f16x8, fb[j & 1][1][nb]);
          acc[x][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[x][nb], 0, 0, 0);
          acc[x][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[x][nb], 0, 0, 0);
          acc[x][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[x][nb], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- the epilogue's operands (conditioner slab | residual stream / skip sum, biases, keep mask), requested HERE, in front of the first
  // operand stage: their (cold) latency overlaps that stage's own and they land behind the K loop.  Unconditional loads: a column >= T
  // lies in the rows' right pad (readable, the value is not used).  They are older than every DMA piece, so the counted waits cover them.
  // Round-3 measurements, batch 1 x 10 s, us per launch (conv + gate / out-projection; timing-only builds, removed since: commit 6735d87): operands loaded after the K loop
  // 15.2 / 12.1;  HERE as register loads 15.1 / 10.7 (9.3 with the pad-free window);  the same 64 x 64 fp32 tile by LDS-DMA in front of the
  // first stage 15.2 / 10.1, by LDS-DMA in the K loop's last two bodies 16.0 / 11.0 (two stages do not cover a cold fetch);  with neither
  // epilogue nor operands 9.1 / 6.2 -- the rest is the K loop's DMA (its MFMAs alone: 7.5 / 4.3).
  float e_a[2][2][4];          // paired: [nb][gate | filter][k]      unpaired: [nb][x][k] = old X / SK
  float e_b[2][4], e_c[2][4];  // unpaired: bias [x][k], next layer's step bias [x][k]
  float e_k[2];                // unpaired: keep [nb]
  FDX_STAMP(0);
  FDX_STAMP_RT0();
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int t = t0 + wc * 32 + nb * 16 + li;
    if constexpr (Epi::kPaired) {
      const int ch0 = mt * 32 + wr * 16 + 4 * kg;
      const float* Pg = epi.P + item * epi.p_bs + t;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        e_a[nb][0][k] = Pg[(long)(ch0 + k) * epi.ldp];
        e_a[nb][1][k] = Pg[(long)(ch0 + k + epi.C) * epi.ldp];
      }
    } else {
      const float* kp = (epi.Yb && epi.keep) ? epi.keep + item * epi.keep_bs + t : epi.bias;
      e_k[nb] = *kp;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int blk0 = mt * 64 + wr * 32 + x * 16, row0 = blk0 + 4 * kg;
        const bool res = blk0 < epi.C;
        const float* RW = (res ? epi.X : epi.SK) + item * epi.bs + (long)(res ? row0 : row0 - epi.C) * epi.ld + t;
#pragma unroll
        for (int k = 0; k < 4; ++k) e_a[nb][x][k] = RW[(long)k * epi.ld];
      }
    }
  }
  if constexpr (!Epi::kPaired) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int blk0 = mt * 64 + wr * 32 + x * 16, row0 = blk0 + 4 * kg;
      const bool use_sb = blk0 < epi.C && epi.Yb;
      const float* sbp = use_sb ? epi.sb + item * epi.sb_bs : epi.bias;
      const long sbs = use_sb ? epi.sb_ld : 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        e_b[x][k] = epi.bias[row0 + k];
        e_c[x][k] = sbp[(long)(row0 + k) * sbs];
      }
    }
  }

  // ---- K loop over the 32-channel blocks, NST stages with compile-time indices (see bf16lds_kernel)
  const int last = a.n_blk - 1;
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
#pragma unroll
    for (int q = 0; q < NP; ++q) piece(q, min(st, last), st);
  wait_landed();
  FDX_STAMP(1);
  auto body = [&](auto S_, int blk) {
    constexpr int S = decltype(S_)::value;
    compute(S, min(blk + NST - 1, last), (S + NST - 1) % NST);
    wait_landed();
  };
  for (int blk = 0; blk < a.n_blk; blk += NST)
    s64_for_each_stage(std::make_integer_sequence<int, NST>{}, [&](auto S_) {
      if (blk + decltype(S_)::value < a.n_blk) body(S_, blk + decltype(S_)::value);
    });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing re-loads must have landed before this workgroup's LDS is released
  FDX_STAMP(2);

  // ---------------------------------------------------------------- epilogue: lane = column li of a 16-column block, rows 4 * kg .. + 3
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int t = t0 + wc * 32 + nb * 16 + li;
    if (t >= a.T) continue;
    if constexpr (Epi::kPaired) {
      const int ch0 = mt * 32 + wr * 16 + 4 * kg;               // this lane's 4 channels: gate rows = acc[0], filter rows = acc[1]
      float z[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        z[k] = EpiGate::gate1(acc[0][nb][k] * epi.acc_scale + e_a[nb][0][k], acc[1][nb][k] * epi.acc_scale + e_a[nb][1][k]);
      bf_store_quad<1>(epi.Zb, item * epi.zb_bs, epi.ldz, ch0, t, z, epi.out_scale);
    } else {
      const float kraw = e_k[nb];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int blk0 = mt * 64 + wr * 32 + x * 16;            // 16 rows entirely on one side of C
        const int row0 = blk0 + 4 * kg;
        const bool res = blk0 < epi.C;
        const long o0 = item * epi.bs + (long)(res ? row0 : row0 - epi.C) * epi.ld + t;
        float* __restrict__ RW = res ? epi.X : epi.SK;
        const bool rd = res || epi.skip_mode == 1 || epi.skip_mode == 2;
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v = acc[x][nb][k] * epi.acc_scale + e_b[x][k];
          if (res) {
            v = div_const(e_a[nb][x][k] + v, 1.41421356237309504880f, 0.70710678118654752440f);
            y[k] = (epi.Yb && (!epi.keep || kraw != 0.f)) ? v + e_c[x][k] : 0.f;
          } else {
            if (rd) v = e_a[nb][x][k] + v;
            if (epi.skip_mode >= 2) v = div_const(v, epi.inv_div, epi.r_inv_div);
          }
          RW[o0 + (long)k * epi.ld] = v;
        }
        if (res && epi.Yb) bf_store_quad<1>(epi.Yb, item * epi.yb_bs, epi.ld, row0, t, y, epi.out_scale);
      }
    }
  }
  FDX_STAMP(5);
  FDX_STAMP_RT1();
}

template <class Epi>
inline hipError_t launch_f16s64(const uint4* Wp, const uint4* Xb, long x_bs, int ld, int C, int dil, int B, int T, int rows, const Epi& epi,
                                hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
  BfArgs a;
  a.Wp = Wp; a.Xb = Xb; a.x_bs = x_bs; a.ld = ld; a.dil = dil; a.T = T;
  a.n_blk = C / 32;
  a.n_mtiles = rows / 64;
  a.tiles_per_item = (T + 63) / 64;
  a.n_tiles_n = B * a.tiles_per_item;
  const int grid = a.n_tiles_n * a.n_mtiles;
  if (grid <= 0) return hipSuccess;
#ifdef FDX_KTRACE
  a.trace = nullptr;
  if (g_trace.buf && g_trace.n < g_trace.max_launches && grid <= g_trace.blocks_cap)
    a.trace = g_trace.buf + (size_t)(g_trace.n++) * g_trace.blocks_cap * 32;
#endif
  // stage count: FDX_F16S_NST (conv: 3 | 4) / FDX_F16S_NST_O (out-projection: 3 | 4; 6 and 8 measured no better) override the defaults (A/B runs)
  static const int nst_c = [] { const char* e = getenv("FDX_F16S_NST"); const int k = e ? atoi(e) : 0; return (k == 3 || k == 4) ? k : kS64StagesConv; }();
  static const int nst_o = [] { const char* e = getenv("FDX_F16S_NST_O"); const int k = e ? atoi(e) : 0; return (k == 3 || k == 4) ? k : kS64StagesOutp; }();
#define FDX_S64_LAUNCH(N)                                                                                        \
  do {                                                                                                             \
    if (ev0) hipExtLaunchKernelGGL((f16s64_kernel<Epi, N>), dim3(grid), dim3(256), 0, s, ev0, ev1, 0, a, epi);     \
    else hipLaunchKernelGGL((f16s64_kernel<Epi, N>), dim3(grid), dim3(256), 0, s, a, epi);                         \
  } while (0)
  if constexpr (Epi::kTaps == 3) {
    if (nst_c == 4) FDX_S64_LAUNCH(4); else FDX_S64_LAUNCH(3);
  } else {
    if (nst_o == 3) FDX_S64_LAUNCH(3); else FDX_S64_LAUNCH(4);
  }
#undef FDX_S64_LAUNCH
  return hipGetLastError();
}

}  // namespace fdx
