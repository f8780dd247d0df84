// convplan.hip.h -- planning / packing / launching of ordinary Conv1d layers on the convgemm kernel family, shared by the
// vocoder translation units (nsf.hip, refinegan.hip).
#pragma once
#include <cstdlib>

#include "common.hip.h"

namespace fdx {

inline PackedW plan_conv(size_t& cur, int rows, int cin, int taps) {
  PackedW p;
  p.RB = rows <= 32 ? 1 : 2;
  p.rows = rows;
  p.cin8 = (cin + 7) / 8;
  p.taps = taps;
  p.n_mtiles = (rows + 32 * p.RB - 1) / (32 * p.RB);
  p.w_off = cur;
  cur += packed_floats(p.n_mtiles, p.RB, p.cin8, p.taps);
  p.b_off = cur;
  cur += (size_t)round_up(rows, 64);
  return p;
}

// Conv1d weight [rows][cin][taps] (+ bias [rows]) -> fragment order
inline void pack_conv1d(float* A, const PackedW& p, const float* w, int rows, int cin, const float* bias) {
  const int R = 32 * p.RB;
  pack_convgemm(A + p.w_off, p.n_mtiles, p.RB, p.cin8, p.taps, [&](int mt, int rb, int i, int c, int tap) -> float {
    const int row = mt * R + rb * 32 + i;
    if (row >= rows || c >= cin) return 0.f;
    return w[((size_t)row * cin + c) * p.taps + tap];
  });
  for (int r = 0; r < rows; ++r) A[p.b_off + r] = bias[r];
}

// Decomposition heuristic: one 64-col tile per wave (no LDS) when that already fills the chip, else 4-wave split-K.
// FDX_NOSPLIT_MIN_WGS overrides the threshold (tuning knob, read once).
inline long no_split_min_wgs() {
  static const long v = [] { const char* e = getenv("FDX_NOSPLIT_MIN_WGS"); return e ? atol(e) : 512L; }();
  return v;
}

template <bool LRELU, class Epi>
inline hipError_t run_conv(const float* arena, const PackedW& p, int B, int T, const float* X, long x_bs, int ldx, int shift0,
                           int dshift, float slope, const Epi& e, hipStream_t s, ProfEvents* prof = nullptr, int prof_kind = -1) {
  ConvGeom g{B, T, p.cin8, p.taps, shift0, dshift, p.n_mtiles};
  const float4* Wp = reinterpret_cast<const float4*>(arena + p.w_off);
  const long wg_nosplit = (long)B * ((T + 255) / 256) * p.n_mtiles;
  if (p.RB == 1) return launch_convgemm<1, false, LRELU, Epi>(g, Wp, X, x_bs, ldx, slope, e, s);
  if (wg_nosplit >= no_split_min_wgs() || p.cin8 * p.taps < 4) {
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // fdx_prof_*: this instantiation is the vocoders' dominant kernel (only their ResBlock convs pass `prof`)
    if (prof) {
      prof->note(prof_kind, "convgemm_kernel<2, false, %d, EpiResblock> (v_mfma_f32_32x32x2_f32; one 64 x 64 tile per wave, no split-K, 64 x 256 per workgroup)",
                 (int)LRELU);
      prof->take(prof_kind, 2.0 * p.rows * (8.0 * p.cin8) * p.taps * (double)B * T, ev0, ev1);
    }
    return launch_convgemm<2, false, LRELU, Epi>(g, Wp, X, x_bs, ldx, slope, e, s, ev0, ev1);
  }
  return launch_convgemm<2, true, LRELU, Epi>(g, Wp, X, x_bs, ldx, slope, e, s);
}

}  // namespace fdx
