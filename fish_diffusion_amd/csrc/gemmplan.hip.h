// gemmplan.hip.h -- plain-GEMM (1x1 conv / Linear) planning, packing and launch helpers shared by the denoisers built on the
// convgemm family beside the WaveNet (convnext.hip, tfdec.hip).
#pragma once
#include "common.hip.h"

namespace fdx {

inline PackedW plan64(size_t& cur, int rows, int cin) {   // plain GEMM, 64-row tiles
  PackedW p;
  p.RB = 2; p.rows = rows; p.cin8 = (cin + 7) / 8; p.taps = 1; p.n_mtiles = (rows + 63) / 64;
  p.w_off = cur; cur += packed_floats(p.n_mtiles, 2, p.cin8, 1);
  p.b_off = cur; cur += (size_t)round_up(rows, 64);
  return p;
}
inline PackedW plan32(size_t& cur, int rows, int cin) {   // 32-row tiles: twice the workgroups for the D-row GEMM with the long K
  PackedW p;
  p.RB = 1; p.rows = rows; p.cin8 = (cin + 7) / 8; p.taps = 1; p.n_mtiles = (rows + 31) / 32;
  p.w_off = cur; cur += packed_floats(p.n_mtiles, 1, p.cin8, 1);
  p.b_off = cur; cur += (size_t)round_up(rows, 64);
  return p;
}


inline void pack_lin(float* A, const PackedW& p, const float* w, int rows, int cin, const float* bias, int row0 = 0) {
  // weight [rows][cin] placed at logical rows row0.. of the packed matrix (used to concatenate per-layer projections)
  const int R = 32 * p.RB;
  const int n_it = p.cin8;
  for (int r = 0; r < rows; ++r) {
    const int row = row0 + r, mt = row / R, rb = (row % R) / 32, i = row % 32;
    for (int c = 0; c < cin; ++c) {
      const int cb = c / 8, half = (c % 8) / 4, j = c % 4;
      A[p.w_off + ((((size_t)mt * n_it + cb) * p.RB + rb) * 64 + half * 32 + i) * 4 + j] = w[(size_t)r * cin + c];
    }
    A[p.b_off + row] = bias ? bias[r] : 0.f;
  }
}


inline EpiBias bias_epi(float* out, long o_bs, int ldo, const float* bias, int M, int act) {
  EpiBias e{};
  e.out = out; e.o_bs = o_bs; e.ldo = ldo; e.bias = bias; e.M = M; e.act = act;
  return e;
}

template <class Epi>
inline hipError_t gemm(const float* A, const PackedW& p, int B, int T, const float* X, long x_bs, int ldx, const Epi& e, hipStream_t s) {
  ConvGeom g{B, T, p.cin8, 1, 0, 0, p.n_mtiles};
  const float4* Wp = reinterpret_cast<const float4*>(A + p.w_off);
  if (p.RB == 1) return launch_convgemm<1, true, false, Epi>(g, Wp, X, x_bs, ldx, 1.f, e, s);
  return launch_convgemm<2, true, false, Epi>(g, Wp, X, x_bs, ldx, 1.f, e, s);
}



}  // namespace fdx
