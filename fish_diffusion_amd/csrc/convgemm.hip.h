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
#include <hip/hip_runtime.h>
#include <stdint.h>

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
  float in_slope;                 // leaky-relu slope applied to the B operand (LRELU instantiations)
};

// ------------------------------------------------------------------------------------------ epilogues
// Each epilogue sees (item b, logical row, column t, value).  `kPaired` epilogues get the values of row
// and row + "pair distance" (same lane, accumulators rb=0 / rb=1): gate/filter, re/im.

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_MISH = 2 };

__device__ __forceinline__ float mish_f(float x) {
  // x * tanh(softplus(x)); F.softplus: beta=1, threshold=20 (wavenet.py:8-10)
  float sp = x > 20.f ? x : log1pf(expf(x));
  return x * tanhf(sp);
}

struct EpiBias {  // out = act(acc + bias[row]); masked columns -> 0; optional out2 = out + sb[row]
  static constexpr bool kPaired = false;
  float* out; long o_bs; int ldo;
  const float* bias;           // [M] or null
  int M, act;
  const uint8_t* mask; int mask_ld;   // [B][mask_ld] bytes, 1 = masked
  float* out2; long o2_bs; int ldo2;  // optional second output
  const float* sb; int sb_ld, sb_bs;  // out2 = v + sb[row*sb_ld + b*sb_bs]
  __device__ __forceinline__ void operator()(int b, int row, int t, float v) const {
    if (row >= M) return;
    if (bias) v += bias[row];
    if (act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (act == ACT_MISH) v = mish_f(v);
    if (mask && mask[(long)b * mask_ld + t]) v = 0.f;
    out[b * o_bs + (long)row * ldo + t] = v;
    if (out2) out2[b * o2_bs + (long)row * ldo2 + t] = v + sb[(long)row * sb_ld + b * sb_bs];
  }
};

struct EpiGate {  // wavenet.py:112-115: y = conv + conditioner (bias folded into P); z = sigmoid(gate)*tanh(filter)
  static constexpr bool kPaired = true;
  float* out; long o_bs; int ldo;
  const float* P; long p_bs; int ldp;  // [B][2C][ldp]: conditioner slab (+ conv bias + conditioner bias)
  int C;
  __device__ __forceinline__ void operator()(int b, int row, int t, float g, float f) const {
    if (row >= C) return;
    const float* p = P + b * p_bs + t;
    g += p[(long)row * ldp];
    f += p[(long)(row + C) * ldp];
    float sg = 1.f / (1.f + expf(-g));
    out[b * o_bs + (long)row * ldo + t] = sg * tanhf(f);
  }
};

struct EpiResSkip {  // wavenet.py:117-120 + the skip sum of :228
  static constexpr bool kPaired = false;
  float* X; float* Y; float* SK; long bs; int ld;   // all [B][C][ld]; Y may be null (last layer)
  const float* bias;                                  // [2C]
  const float* sb; int sb_ld, sb_bs;                  // next layer's diffusion projection, [C][sb_ld]
  int C, skip_mode;                                   // 0 first (=), 1 middle (+=), 2 last ((+=)/sqrt(L)); 3 = first and last
  float inv_div;                                      // sqrt(n_layers)
  __device__ __forceinline__ void operator()(int b, int row, int t, float v) const {
    if (row >= 2 * C) return;
    v += bias[row];
    if (row < C) {
      long o = b * bs + (long)row * ld + t;
      float xn = (X[o] + v) / 1.41421356237309504880f;
      X[o] = xn;
      if (Y) Y[o] = xn + sb[(long)row * sb_ld + b * sb_bs];
    } else {
      long o = b * bs + (long)(row - C) * ld + t;
      float s = v;
      if (skip_mode == 1 || skip_mode == 2) s = SK[o] + v;
      if (skip_mode >= 2) s = s / inv_div;
      SK[o] = s;
    }
  }
};

struct EpiResblock {  // models.py:103-110 conv2: x = xt + x; plus the MRF mean of :426-432
  static constexpr bool kPaired = false;
  float* out; const float* resid; long bs; int ld;
  const float* bias; int M;
  int mode;     // 0: out = v    1: out += v    2: out = (out + v) / div
  float div;
  __device__ __forceinline__ void operator()(int b, int row, int t, float v) const {
    if (row >= M) return;
    long o = b * bs + (long)row * ld + t;
    v += bias[row];
    if (resid) v += resid[o];
    if (mode == 1) v = out[o] + v;
    else if (mode == 2) v = (out[o] + v) / div;
    out[o] = v;
  }
};

struct EpiUps {  // polyphase ConvTranspose1d (models.py:421): logical row = phase*Cout + co
  static constexpr bool kPaired = false;
  float* out; long o_bs; int ldo;
  const float* bias; int Cout, stride, Lout;
  __device__ __forceinline__ void operator()(int b, int row, int t, float v) const {
    int ph = row / Cout;
    if (ph >= stride) return;
    int co = row - ph * Cout;
    int n = t * stride + ph;
    if (n >= Lout) return;
    out[b * o_bs + (long)co * ldo + n] = v + bias[co];
  }
};

struct EpiMag {  // pitch_adjustable_mel.py:83-92: sqrt(re^2 + im^2 + 1e-9) [* win_size / win_new]
  static constexpr bool kPaired = true;
  float* out; long o_bs; int ldo;
  int n_bins, n_rows;   // rows in [n_bins, n_rows) are written as 0 (zero-padded bins)
  float mul, div;       // 0 => no rescale
  __device__ __forceinline__ void operator()(int b, int row, int t, float re, float im) const {
    if (row >= n_rows) return;
    float v = 0.f;
    if (row < n_bins) {
      v = sqrtf(re * re + im * im + 1e-9f);
      if (mul != 0.f) v = v * mul / div;
    }
    out[b * o_bs + (long)row * ldo + t] = v;
  }
};

struct EpiLogMel {  // audio.py:11-18 + nsf_hifigan.py:104-105
  static constexpr bool kPaired = false;
  float* out; long o_bs; int ldo; int M, log_mode;
  __device__ __forceinline__ void operator()(int b, int row, int t, float v) const {
    if (row >= M) return;
    if (log_mode != 0) {
      v = logf(fmaxf(v, 1e-5f));
      if (log_mode == 2) v = 0.434294f * v;
    }
    out[b * o_bs + (long)row * ldo + t] = v;
  }
};

// ------------------------------------------------------------------------------------------ kernel
// Accumulator element r of a 32x32 tile sits at row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31.
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int RB, bool SPLITK, bool LRELU, class Epi>
__global__ __launch_bounds__(256) void convgemm_kernel(ConvArgs a, Epi epi) {
  static_assert(!Epi::kPaired || RB == 2, "paired epilogues need both row blocks");
  constexpr int NB = 2;                       // two 32-column blocks per wave tile
  constexpr int Q = RB * NB * 16;             // accumulator registers per lane
  __shared__ float red[SPLITK ? 4 * Q * kWave : 1];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, li = lane & 31;

  // ---- XCD-aware logical tile id (block b runs on XCD b % 8; give each XCD a contiguous chunk)
  const int G = gridDim.x, bid = blockIdx.x;
  const int q8 = G >> 3, r8 = G & 7, xcd = bid & 7;
  const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int mt = L / a.n_tiles_n;
  const int nt = L - mt * a.n_tiles_n;
  const int item = nt / a.tiles_per_item;
  const int tile_in_item = nt - item * a.tiles_per_item;
  constexpr int COLS = SPLITK ? 64 : 256;
  const int t0 = tile_in_item * COLS + (SPLITK ? 0 : wave * 64);

  int it_begin = 0, it_end = a.n_it;
  if (SPLITK) {
    const int per = (a.n_it + 3) >> 2;
    it_begin = wave * per;
    it_end = min(a.n_it, it_begin + per);
  }

  f32x16 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.f;

  const bool active = SPLITK ? true : (t0 < a.T);   // whole-wave overhang tiles skip the K loop
  if (active && it_begin < it_end) {
    const float4* Ap = a.Wp + ((long)mt * a.n_it + it_begin) * (RB * 64) + lane;
    const float* Xw = a.X + item * a.x_bstride + (long)(half * 4) * a.ldx + t0 + li;
    int cb = it_begin / a.taps;
    int tap = it_begin - cb * a.taps;

    float4 a_cur[RB], a_nxt[RB];
    float b_cur[4][NB], b_nxt[4][NB];

    auto load = [&](float4(&av)[RB], float(&bv)[4][NB], int cbi, int tapi, const float4* ap) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) av[rb] = ap[rb * 64];
      const float* xp = Xw + (long)(cbi * 8) * a.ldx + (a.shift0 + tapi * a.dshift);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[j][nb] = xp[(long)j * a.ldx + nb * 32];
    };

    load(a_cur, b_cur, cb, tap, Ap);
    // Make the prologue loads land before the loop: otherwise hipcc's waitcnt pass merges "pending
    // prologue load" into the loop header state and puts vmcnt waits for the NEXT tile's loads in front
    // of the current tile's MFMAs (measured in the .s: vmcnt(7..0) ladder inside the MFMA block).
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
      asm volatile("" : "+v"(a_cur[rb].x), "+v"(a_cur[rb].y), "+v"(a_cur[rb].z), "+v"(a_cur[rb].w));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(b_cur[j][nb]));
    for (int it = it_begin; it < it_end; ++it) {
      Ap += RB * 64;
      if (++tap == a.taps) { tap = 0; ++cb; }
      if (it + 1 < it_end) load(a_nxt, b_nxt, cb, tap, Ap);
      if (LRELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            float v = b_cur[j][nb];
            b_cur[j][nb] = v > 0.f ? v : v * a.in_slope;
          }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const float av = j == 0 ? a_cur[rb].x : j == 1 ? a_cur[rb].y : j == 2 ? a_cur[rb].z : a_cur[rb].w;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_cur[j][nb], acc[rb][nb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a_cur[rb] = a_nxt[rb];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b_cur[j][nb] = b_nxt[j][nb];
    }
  }

  const int row_base = mt * (Epi::kPaired ? 32 : 32 * RB);

  if (SPLITK) {
    // ---- cross-wave K reduction through LDS, fixed summation order (deterministic)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          red[(wave * Q + (rb * NB + nb) * 16 + r) * kWave + lane] = acc[rb][nb][r];
    __syncthreads();
    auto rsum = [&](int q) {
      return ((red[(0 * Q + q) * kWave + lane] + red[(1 * Q + q) * kWave + lane]) +
              red[(2 * Q + q) * kWave + lane]) + red[(3 * Q + q) * kWave + lane];
    };
    if constexpr (Epi::kPaired) {
      const int nb = wave >> 1, r0 = (wave & 1) * 8;
      const int t = t0 + nb * 32 + li;
#pragma unroll
      for (int r = r0; r < r0 + 8; ++r) {
        float g = rsum((0 * NB + nb) * 16 + r), f = rsum((1 * NB + nb) * 16 + r);
        if (t < a.T) epi(item, row_base + acc_row(r, half), t, g, f);
      }
    } else if constexpr (RB == 2) {
      const int rb = wave >> 1, nb = wave & 1;
      const int t = t0 + nb * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = rsum((rb * NB + nb) * 16 + r);
        if (t < a.T) epi(item, row_base + rb * 32 + acc_row(r, half), t, v);
      }
    } else {
      const int nb = wave >> 1, r0 = (wave & 1) * 8;
      const int t = t0 + nb * 32 + li;
#pragma unroll
      for (int r = r0; r < r0 + 8; ++r) {
        float v = rsum(nb * 16 + r);
        if (t < a.T) epi(item, row_base + acc_row(r, half), t, v);
      }
    }
  } else {
    if (!active) return;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int t = t0 + nb * 32 + li;
      if (t >= a.T) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (Epi::kPaired) {
          epi(item, row_base + acc_row(r, half), t, acc[0][nb][r], acc[1][nb][r]);
        } else {
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) epi(item, row_base + rb * 32 + acc_row(r, half), t, acc[rb][nb][r]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ host launch
struct ConvGeom {   // everything the launcher needs besides pointers
  int B, T;         // items, valid columns per item
  int cin8;         // Cin / 8 (padded)
  int taps, shift0, dshift;
  int n_mtiles;     // row tiles of 32*RB logical rows (32 pairs for paired epilogues)
};

template <int RB, bool SPLITK, bool LRELU, class Epi>
inline hipError_t launch_convgemm(const ConvGeom& g, const float4* Wp, const float* X, long x_bstride, int ldx,
                                  float in_slope, const Epi& epi, hipStream_t s) {
  ConvArgs a;
  a.Wp = Wp; a.X = X; a.x_bstride = x_bstride; a.ldx = ldx;
  a.n_it = g.cin8 * g.taps; a.taps = g.taps; a.shift0 = g.shift0; a.dshift = g.dshift;
  a.T = g.T;
  const int cols = SPLITK ? 64 : 256;
  a.tiles_per_item = (g.T + cols - 1) / cols;
  a.n_tiles_n = g.B * a.tiles_per_item;
  a.n_mtiles = g.n_mtiles;
  a.in_slope = in_slope;
  const int grid = a.n_tiles_n * a.n_mtiles;
  if (grid <= 0) return hipSuccess;
  hipLaunchKernelGGL((convgemm_kernel<RB, SPLITK, LRELU, Epi>), dim3(grid), dim3(256), 0, s, a, epi);
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
