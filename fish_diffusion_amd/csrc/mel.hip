// mel.hip -- STFT -> magnitude -> mel filterbank -> log  (fish_diffusion/utils/pitch_adjustable_mel.py:33-96,
// utils/audio.py:11-18, nsf_hifigan.py:101-107).
//
// The key-shift variant needs a DFT of arbitrary length (round(2048 * 2^(ks/12)), e.g. 2170), so the transform is
// expressed as a dense [2*bins x n_fft] real GEMM on the fp32 MFMA kernel: one "frames" kernel applies reflect padding +
// Hann window and lays frames out as [n_fft][T]; the DFT GEMM's paired epilogue (cos rows / -sin rows in the same lane)
// produces sqrt(re^2+im^2+1e-9) directly; the mel GEMM's epilogue applies clamp+log.  7 GFLOP per 10 s utterance --
// <0.1 % of the hot path -- and exact for every n_fft.
#include "common.hip.h"
#include "elementwise.hip.h"

#include <cmath>

using namespace fdx;

// ================================================================================================ filterbank (host)
// librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm="slaney") -- librosa 0.9.1 algorithm, float64
// arithmetic with the float32 stores of the original (weights array is float32, scaled in place by float64 enorm).
static double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}
static void np_linspace(double a, double b, int n, std::vector<double>& out) {
  out.resize(n);
  const double step = (b - a) / (n - 1);
  for (int i = 0; i < n; ++i) out[i] = i * step + a;
  out[n - 1] = b;
}

static int mel_validate(const fdx_mel_desc* d) {
  if (!d) return fail(nullptr, FDX_E_ARG, "null mel desc");
  if (d->sample_rate <= 0 || d->n_fft < 16 || d->win_size <= 0 || d->win_size > d->n_fft || d->hop <= 0 || d->n_mels <= 0)
    return fail(nullptr, FDX_E_ARG, "bad mel geometry");
  if (!(d->f_min >= 0.f) || !(d->f_max > d->f_min)) return fail(nullptr, FDX_E_ARG, "bad f_min/f_max");
  return FDX_OK;
}

extern "C" int fdx_mel_filterbank(const fdx_mel_desc* d, float* out) {
  if (mel_validate(d) || !out) return FDX_E_ARG;
  const int n_bins = 1 + d->n_fft / 2, n_mels = d->n_mels;
  std::vector<double> fftfreqs, mel_pts, mel_f(n_mels + 2);
  np_linspace(0.0, (double)d->sample_rate / 2, n_bins, fftfreqs);
  np_linspace(hz_to_mel(d->f_min), hz_to_mel(d->f_max), n_mels + 2, mel_pts);
  for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(mel_pts[i]);
  for (int i = 0; i < n_mels; ++i) {
    const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
    const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
    for (int j = 0; j < n_bins; ++j) {
      const double lower = -(mel_f[i] - fftfreqs[j]) / fd0;
      const double upper = (mel_f[i + 2] - fftfreqs[j]) / fd1;
      const float w32 = (float)std::fmax(0.0, std::fmin(lower, upper));
      out[(size_t)i * n_bins + j] = (float)((double)w32 * enorm);
    }
  }
  return FDX_OK;
}

struct StftGeom { int n_fft, win, hop, pad, T, bins; };
static int stft_geom(const fdx_mel_desc& d, int N, float key_shift, float speed, StftGeom& g) {
  // pitch_adjustable_mel.py:34-37: np.round (half to even) of the scaled sizes
  const double factor = std::pow(2.0, (double)key_shift / 12.0);
  g.n_fft = (int)std::nearbyint(d.n_fft * factor);
  g.win = (int)std::nearbyint(d.win_size * factor);
  g.hop = (int)std::nearbyint(d.hop * (double)speed);
  if (g.n_fft < 2 || g.win < 1 || g.win > g.n_fft || g.hop < 1) return FDX_E_ARG;
  g.pad = (g.win - g.hop) / 2;   // int((win - hop) / 2): truncation toward zero
  if (g.win < g.hop) g.pad = -((g.hop - g.win) / 2);
  if (g.pad < 0 || g.pad >= N) return FDX_E_ARG;   // reflect padding needs pad < N
  const int Lp = N + 2 * g.pad;
  if (Lp < g.n_fft) return FDX_E_ARG;
  g.T = 1 + (Lp - g.n_fft) / g.hop;
  g.bins = 1 + g.n_fft / 2;
  return FDX_OK;
}

extern "C" int fdx_mel_num_frames(const fdx_mel_desc* d, int N, float key_shift, float speed, int* T) {
  if (mel_validate(d) || !T) return FDX_E_ARG;
  StftGeom g;
  if (stft_geom(*d, N, key_shift, speed, g)) return fail(nullptr, FDX_E_ARG, "input of %d samples is too short for this STFT geometry", N);
  *T = g.T;
  return FDX_OK;
}

extern "C" int fdx_mel_config(fdx_handle h, const fdx_mel_desc* d) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (mel_validate(d)) { h->err = g_last_error; return FDX_E_ARG; }
  FDX_HIP(h, hipSetDevice(h->device));
  const int n_bins = 1 + d->n_fft / 2;
  std::vector<float> fb((size_t)d->n_mels * n_bins);
  if (int rc = fdx_mel_filterbank(d, fb.data())) return rc;
  const int cin8 = (n_bins + 7) / 8, n_mtiles = (d->n_mels + 63) / 64;
  std::vector<float> packed(packed_floats(n_mtiles, 2, cin8, 1));
  pack_convgemm(packed.data(), n_mtiles, 2, cin8, 1, [&](int mt, int rb, int i, int c, int) -> float {
    const int row = mt * 64 + rb * 32 + i;
    return (row < d->n_mels && c < n_bins) ? fb[(size_t)row * n_bins + c] : 0.f;
  });
  FDX_HIP(h, h->mel_basis_packed.ensure(packed.size() * 4, false, nullptr));
  FDX_HIP(h, hipMemcpy(h->mel_basis_packed.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  h->md = *d;
  h->mel_ok = true;
  fdx_mel_free_tables(h);        // tables of another base geometry (bins cropped at the base resolution) are stale
  return FDX_OK;
}

void fdx_mel_free_tables(fdx_ctx* h) {
  for (auto* t : h->mel_tables) {
    if (t->host) (void)hipHostFree(t->host);
    delete t;
  }
  h->mel_tables.clear();
}

extern "C" int fdx_mel_stats(fdx_handle h, long* table_builds, long* stream_syncs, int* cached) {
  if (!h) return FDX_E_ARG;
  if (table_builds) *table_builds = h->mel_builds;
  if (stream_syncs) *stream_syncs = h->mel_syncs;
  if (cached) *cached = (int)h->mel_tables.size();
  return FDX_OK;
}

// ================================================================================================ device kernels
// FR[b][k][t] = window[k] * reflect_pad(wav[b])[t*hop + k]   (pitch_adjustable_mel.py:61-82; torch.stft center=False)
__global__ void k_frames(float* __restrict__ FR, long fr_bs, int ldf, const float* __restrict__ wav, int N,
                         const float* __restrict__ window, int n_fft, int hop, int pad, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int k = blockIdx.y, b = blockIdx.z;
  int i = t * hop + k - pad;           // index into the unpadded signal
  if (i < 0) i = -i;                   // reflect (no edge repeat)
  if (i >= N) i = 2 * (N - 1) - i;
  FR[b * fr_bs + (long)k * ldf + t] = window[k] * wav[(long)b * N + i];
}

// ================================================================================================ forward
// The DFT matrix (packed for the GEMM) and the Hann window of one STFT geometry, from the handle's cache.  A new geometry is built on the host
// into PINNED memory owned by the cache entry and uploaded with hipMemcpyAsync on the caller's stream: no stream synchronisation, and a caller
// alternating key shifts (the reference's augmentation path) pays the build once per shift.  Only evicting the least recently used of
// kMelTables geometries waits for the stream (its arena may still be read by queued launches) -- counted in mel_syncs.
static int ensure_dft(fdx_ctx* h, const StftGeom& g, hipStream_t s, const fdx_ctx::MelTable*& out) {
  for (auto* t : h->mel_tables)
    if (t->n_fft == g.n_fft && t->win == g.win) { t->last_use = ++h->mel_clock; out = t; return FDX_OK; }
  const int bins_out = std::min(g.bins, 1 + h->md.n_fft / 2);   // bins beyond the base resolution are cropped (:91)
  const int n_mtiles = (bins_out + 31) / 32, cin8 = (g.n_fft + 7) / 8;
  const double w0 = 2.0 * 3.14159265358979323846 / g.n_fft;
  const size_t dft_floats = packed_floats(n_mtiles, 2, cin8, 1), win_floats = (size_t)round_up(g.n_fft, 8);
  if ((int)h->mel_tables.size() >= fdx_ctx::kMelTables) {
    size_t lru = 0;
    for (size_t i = 1; i < h->mel_tables.size(); ++i) if (h->mel_tables[i]->last_use < h->mel_tables[lru]->last_use) lru = i;
    FDX_HIP(h, hipStreamSynchronize(s));   // the evicted arena's last readers
    ++h->mel_syncs;
    if (h->mel_tables[lru]->host) (void)hipHostFree(h->mel_tables[lru]->host);
    delete h->mel_tables[lru];
    h->mel_tables.erase(h->mel_tables.begin() + lru);
  }
  auto* t = new fdx_ctx::MelTable();
  t->n_fft = g.n_fft; t->win = g.win; t->dft_floats = dft_floats; t->floats = dft_floats + win_floats;
  hipError_t e = hipHostMalloc(&t->host, t->floats * sizeof(float), hipHostMallocDefault);
  if (e == hipSuccess) e = t->dev.ensure(t->floats * sizeof(float), false, s);
  if (e != hipSuccess) {
    if (t->host) (void)hipHostFree(t->host);
    delete t;
    return fail(h, FDX_E_HIP, "mel tables for n_fft = %d: %s", g.n_fft, hipGetErrorString(e));
  }
  float* packed = static_cast<float*>(t->host);
  pack_convgemm(packed, n_mtiles, 2, cin8, 1, [&](int mt, int rb, int i, int c, int) -> float {
    const int f = mt * 32 + i;
    if (f >= bins_out || c >= g.n_fft) return 0.f;
    const long m = ((long)f * c) % g.n_fft;   // exact angle reduction
    return rb == 0 ? (float)std::cos(w0 * m) : (float)-std::sin(w0 * m);
  });
  // torch.hann_window(win) (periodic), fp32 arithmetic: cos(n * (2 pi / win)) * -0.5 + 0.5; centred in n_fft when shorter
  float* window = packed + dft_floats;
  for (size_t n = 0; n < win_floats; ++n) window[n] = 0.f;
  const float step = (float)(3.14159265358979323846 * 2 / g.win);
  const int left = (g.n_fft - g.win) / 2;
  for (int n = 0; n < g.win; ++n) window[left + n] = std::cos((float)n * step) * -0.5f + 0.5f;
  FDX_HIP(h, hipMemcpyAsync(t->dev.p, t->host, t->floats * sizeof(float), hipMemcpyHostToDevice, s));   // (pinned source owned by the entry)
  t->last_use = ++h->mel_clock;
  ++h->mel_builds;
  h->mel_tables.push_back(t);
  out = t;
  return FDX_OK;
}

extern "C" int fdx_mel_forward(fdx_handle h, const float* wav, int B, int N, float key_shift, float speed, int log_mode,
                               float* mel, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!h->mel_ok) return fail(h, FDX_E_STATE, "fdx_mel_forward: call fdx_mel_config first");
  if (!wav || !mel || B <= 0 || N <= 0) return fail(h, FDX_E_ARG, "fdx_mel_forward: bad arguments");
  if (log_mode < 0 || log_mode > 2) return fail(h, FDX_E_ARG, "fdx_mel_forward: bad log_mode");
  hipStream_t s = as_stream(st);
  FDX_HIP(h, hipSetDevice(h->device));
  const auto& d = h->md;
  StftGeom g;
  if (stft_geom(d, N, key_shift, speed, g)) return fail(h, FDX_E_ARG, "input of %d samples is too short for this STFT geometry", N);
  const fdx_ctx::MelTable* tab = nullptr;
  if (int rc = ensure_dft(h, g, s, tab)) return rc;
  const int T = g.T, ldt = padded_ld(T, 64);
  const int Kp = round_up(g.n_fft, 8);
  const int base_bins = 1 + d.n_fft / 2, rows_spec = round_up(base_bins, 8);
  const int bins_out = std::min(g.bins, base_bins);
  FDX_HIP(h, h->frames.ensure((size_t)B * Kp * ldt * 4, true, s));
  FDX_HIP(h, h->spec.ensure((size_t)B * rows_spec * ldt * 4, true, s));
  const int cin8 = Kp / 8;
  const float* window = tab->dev.f() + tab->dft_floats;
  hipLaunchKernelGGL(k_frames, dim3((T + 255) / 256, g.n_fft, B), dim3(256), 0, s, h->frames.f() + kHalo, (long)Kp * ldt, ldt, wav, N,
                     window, g.n_fft, g.hop, g.pad, T);
  {
    EpiMag e{};
    e.out = h->spec.f() + kHalo; e.o_bs = (long)rows_spec * ldt; e.ldo = ldt; e.n_bins = bins_out; e.n_rows = base_bins;
    if (key_shift != 0.f) { e.mul = (float)d.win_size; e.div = (float)g.win; }   // spec * win_size / win_size_new (:91)
    ConvGeom cg{B, T, cin8, 1, 0, 0, (bins_out + 31) / 32};
    FDX_HIP(h, (launch_convgemm<2, true, false, EpiMag>(cg, reinterpret_cast<const float4*>(tab->dev.p), h->frames.f() + kHalo,
                                                          (long)Kp * ldt, ldt, 1.f, e, s)));
  }
  {
    EpiLogMel e{};
    e.out = mel; e.o_bs = (long)d.n_mels * T; e.ldo = T; e.M = d.n_mels; e.log_mode = log_mode;
    ConvGeom cg{B, T, rows_spec / 8, 1, 0, 0, (d.n_mels + 63) / 64};
    FDX_HIP(h, (launch_convgemm<2, true, false, EpiLogMel>(cg, reinterpret_cast<const float4*>(h->mel_basis_packed.p),
                                                             h->spec.f() + kHalo, (long)rows_spec * ldt, ldt, 1.f, e, s)));
  }
  return FDX_OK;
}
