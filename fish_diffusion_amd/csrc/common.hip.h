// common.hip.h -- context, error plumbing and device-buffer helpers shared by the translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fishdx.h"
#include "convgemm.hip.h"

namespace fdx {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// Row pitch (floats) of a padded activation row holding T valid columns: kHalo zeros, T values, zeros up to a
// multiple of `cols` (tile overhang) plus kHalo more.  Always a multiple of 32 floats (128 B).
inline int padded_ld(int T, int cols = 256) { return kHalo + round_up(T, cols) + kHalo; }

// Allocation generation of the context whose API call is running on this thread (GenScope below): bumped whenever one of ITS
// DevBufs (re)allocates.  Recorded hipGraphs bake buffer addresses in, so their keys include the context's counter.  Per context
// and thread-local on purpose: a process-global counter would race between handles driven from different threads (the handles
// are only serialised per handle) and would make one handle's reallocation throw away every other handle's graphs.
inline thread_local uint64_t* tl_alloc_gen = nullptr;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool in_graphs = true;   // false: per-call scratch no recorded sampler graph ever reads -- its (re)allocation does not re-key the recordings
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  // grow-only; returns hipSuccess.  Contents are zeroed whenever `zero` is set (geometry change).
  hipError_t ensure(size_t bytes, bool zero, hipStream_t s) {
    if (bytes > cap) {
      release();
      hipError_t e = hipMalloc(&p, bytes);
      if (e != hipSuccess) { p = nullptr; return e; }
      cap = bytes;
      if (tl_alloc_gen && in_graphs) ++*tl_alloc_gen;
    }
    if (zero && bytes) return hipMemsetAsync(p, 0, bytes, s);
    return hipSuccess;
  }
  float* f() const { return reinterpret_cast<float*>(p); }
};

// One packed conv/GEMM weight: offsets (in floats) into the arena.
struct PackedW {
  size_t w_off = 0;     // fragment-ordered weights
  size_t b_off = 0;     // bias (logical row order), always present (zeros when the layer has none)
  int n_mtiles = 0, RB = 2, cin8 = 0, taps = 1;
  int rows = 0;         // logical rows (bias length)
};

struct WavenetLayout {
  PackedW in_proj, mlp0, mlp2, dproj, cond, skip_proj, out_proj;
  std::vector<PackedW> conv, outp;   // per layer
  std::vector<int> dil;
  size_t total_floats = 0;
};

struct NsfStage {
  PackedW ups;            // polyphase transposed conv; rows = stride*Cout
  int ups_shift0 = 0;     // tap j reads input column q + ups_shift0 + j
  size_t nc_w = 0, nc_b = 0; int nc_k = 1, nc_stride = 1, nc_pad = 0;   // noise conv (VALU kernel), raw layout
  int cin = 0, cout = 0, stride = 1, ksize = 1;
  std::vector<PackedW> c1, c2;   // [n_resblock_kernels * n_dil]; c2 empty for ResBlock2
  // small-channel stages (C = 16 / 32, ResBlock1 with three dilations): the same weights once more in the fused kernel's order
  // (resblock_fused.hip.h): per ResBlock 6 convs x KS*C*C floats at fw[j], 6 x C biases at fb[j]
  bool fused = false;
  std::vector<size_t> fw, fb;
};
struct NsfLayout {
  size_t src_w = 0, src_b = 0;        // m_source.l_linear
  PackedW conv_pre;
  std::vector<NsfStage> stages;
  size_t post_w = 0, post_b = 0; int post_c = 0;   // conv_post (VALU kernel), raw [1][C][7]
  size_t total_floats = 0;
};

// Which launches fdx_prof_* times (fdx_prof_select): one kernel family at a time
enum { PROF_WN_CONVGATE = 0,    // WaveNet: dilated conv k=3 + gate (convgemm16s_kernel<EpiGate16S> / ...)
       PROF_WN_OUTPROJ = 1,     // WaveNet: out-projection + residual / skip
       PROF_NSF_RESBLOCK = 2,   // NSF-HiFiGAN: the ResBlock convs on convgemm_kernel<2,false,PRE_LRELU,EpiResblock> (C >= 64 stages)
       PROF_RG_RESBLOCK = 3,    // RefineGAN: the same instantiation inside its ResBlocks (down path + ParallelResBlocks)
       PROF_CN_PWCONV1 = 4,     // ConvNext: pwconv1 (LayerNorm folded in, GELU epilogue)
       PROF_TD_ATTN = 5,        // TransformerDecoder: k_attn_qs (fp32 flash attention, self + cross; the combine kernel is not in the interval)
       PROF_KINDS = 6 };
struct ProfEvents {
  bool on = false;
  int kind = PROF_WN_CONVGATE;
  int stride = 1;          // record every stride-th launch of the selected kernel
  long seen = 0;
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
  double flops_total = 0;  // algorithmic FLOPs of the recorded launches
  char label[256] = {0};   // which kernel instantiation / tile shape the RECORDED launches ran (fdx_prof_label)
  char pending[224] = {0}; // what the next launch of the selected family is (note()); becomes `label` when take() samples it
  bool mixed = false;
  void note(int k, const char* fmt, ...) __attribute__((format(printf, 3, 4))) {
    if (!on || k != kind) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(pending, sizeof pending, fmt, ap);
    va_end(ap);
  }
  // events for this launch, or false (not selected / not sampled).  Event-creation errors simply skip the launch.
  bool take(int k, double flops, hipEvent_t& ev0, hipEvent_t& ev1) {
    if (!on || k != kind || (seen++ % stride) != 0) return false;
    if (used == start.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess) return false;
      if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return false; }
      start.push_back(a); stop.push_back(b);
    }
    ev0 = start[used]; ev1 = stop[used]; ++used;
    flops_total += flops;
    // the label names what was SAMPLED: the first sampled launch's instantiation, flagged once if a later sampled launch ran another
    // one (micro-batches of different size may pick different tile shapes / kernel families)
    if (!label[0]) snprintf(label, sizeof label, "%s", pending);
    else if (!mixed && pending[0] && strncmp(label, pending, strlen(pending)) != 0) {
      mixed = true;
      const size_t n = strlen(label);
      snprintf(label + n, sizeof label - n, " [+ other instantiations among the sampled launches]");
    }
    return true;
  }
};

}  // namespace fdx

namespace fdx {
constexpr int kHeads = 8;   // nn.TransformerDecoderLayer(nhead=8), convnext.py:300 (declayer.hip.h)
// key splits of the query-split attention kernel (declayer.hip.h): enough workgroups for ~7/8 of the 256 CUs, never finer than one 32-key unit.
// Host and device: an item of an exact-ragged row picks its own split from its own length; the sampler-graph key needs the largest one.
__host__ __device__ inline int attn_ksplit_of(int B, int Tq, int Tk, int forced) {
  const int units = (Tk + 31) / 32;
  const long base = (long)B * kHeads * ((Tq + 127) / 128);
  int ks = forced > 0 ? forced : (int)((224 + base - 1) / base);
  ks = ks < 8 ? ks : 8;
  ks = ks < units ? ks : units;
  return ks > 1 ? ks : 1;
}
}  // namespace fdx

struct fdx_ctx {
  int device = 0;
  std::string err;

  // ---- wavenet
  bool wn_ok = false;
  fdx_wavenet_desc wd{};
  fdx::WavenetLayout wl;
  const float* wn_arena = nullptr;
  int B = 0, T = 0, ld = 0;            // prepared geometry
  bool prepared = false;
  fdx::DevBuf xin, X, Y, Z, SK, H, EPS, P, condp, condraw, P2;
  fdx::DevBuf wn_nr2;                    // dilated-conv weights in the NR = 2 fragment order (convgemm16s.hip.h), derived at attach
  std::vector<size_t> wn_nr2_off;        // per layer, in floats
  int conv_shape_nr = 4, conv_shape_nm = 4;   // tile shape of the dilated conv + gate for the prepared geometry
  fdx::DevBuf wn_outp16;                 // out-projection weights in the 16x16x4 orders (NR = 4 and NR = 2), derived at attach
  std::vector<size_t> wn_outp16_off4, wn_outp16_off2, wn_outp16_off1;
  int outp_shape_nr = 4, outp_shape_nm = 4;   // tile shape of the out-projection for the prepared geometry
  const void* wn_arena_bf16 = nullptr;   // opt-in bf16 storage mode: residual-block weights as bf16 fragments (wavenet.hip)
  fdx::DevBuf Yb, Zb;                    // ... and the two GEMM operands in C8-blocked bf16
  fdx::DevBuf wn_bf16_lds;               // the same bf16 weights in the LDS-tiled kernels' order (bf16lds.hip.h), derived at bf16 attach
  bool wn_bf16_lds_ok = false;
  int bf16_B = 0, bf16_T = 0;            // geometry Yb / Zb were last zeroed for
  fdx::DevBuf wn_f16s;                   // fp16-split mode: {hi, lo} fp16 weights of the two residual-block GEMMs in LDS order (derived from the fp32 arena)
  bool wn_f16s_ok = false;
  fdx::DevBuf wn_f16s64;                 // ... and in the small-tile kernel's order (f16s64.hip.h), when FDX_F16S_SMALL is set
  bool wn_f16s64_ok = false;
  fdx::DevBuf Yh, Zh;                    // ... and the two GEMM operands as blocked {hi, lo} fp16
  int f16s_B = 0, f16s_T = 0;
  bool cond_masked = false; int condraw_ld = 0;
  fdx::DevBuf tdev, E, Hm, S0, S;      // step-embedding pipeline; ldn below
  int n_emb = 0, ldn = 0;
  // ---- sampler body as a cached hipGraph (wavenet.hip: fdx_sampler_run)
  struct GraphEntry { uint64_t key; hipGraphExec_t exec; uint64_t last_use; };
  std::vector<GraphEntry> graphs;   // LRU, at most graph_cap entries (FDX_GRAPH_CACHE overrides)
  int graph_cap = 48;
  uint64_t graph_clock = 0;
  long graph_captures = 0, graph_launches = 0;
  hipStream_t cap_stream = nullptr;
  bool use_graphs = true;
  // ---- sampler state (padded [B][M][ld])
  fdx::DevBuf sx, sxt, sbase, sm[2], shist[4], seps2, snoise, maskbuf;
  std::vector<float> ts_host;
  // exact-mask runs (fdx_sampler_run_ragged): 1 / 0 per frame in the padded row layout
  fdx::DevBuf keepbuf;
  const float* ragged_keep = nullptr;   // non-null only while such a run is being enqueued / recorded
  // item layout of an exact-ragged row (fdx_sampler_set_items): what the attention-based denoisers need beside the hole mask
  std::vector<int> items;               // host: {offset, length} per item; empty = dense batches
  fdx::DevBuf items_dev;                // int4 per item {offset, length, 0, 0}
  fdx::DevBuf pidx_dev;                 // int per column of the row: position inside its item (0 in holes)
  uint64_t items_hash = 0;              // part of the sampler-graph key (grids depend on the layout)
  int items_max_len = 0, items_T = 0;
  int n_items() const { return (int)items.size() / 2; }

  // ---- nsf
  bool nsf_ok = false;
  fdx_nsf_desc nd{};
  fdx::NsfLayout nl;
  const float* nsf_arena = nullptr;
  int vB = 0, vT = 0;
  fdx::DevBuf vmel, vpre, vf0up, vrad, vscan, vhar, vnoise;
  std::vector<fdx::DevBuf> vU, vR, vTm, vXS;   // per stage
  fdx::DevBuf scan_part;

  // ---- refinegan (opaque: refinegan.hip owns the type)
  void* rg = nullptr;
  // ---- convnext denoiser (opaque: convnext.hip owns the type); den_kind says which denoiser the sampler loop drives
  void* cn = nullptr;
  void* td = nullptr;    // transformer-decoder denoiser (tfdec.hip)
  int den_kind = 0;      // 0 = WaveNet (wavenet.hip), 1 = ConvNext (convnext.hip), 2 = TransformerDecoder (tfdec.hip)
  int den_M = 0;         // mel channels of the prepared denoiser

  // ---- mel
  bool mel_ok = false;
  fdx_mel_desc md{};
  fdx::DevBuf mel_basis_packed, frames, spec;
  // DFT matrix + Hann window per STFT geometry (n_fft, win): a key shift changes both (pitch_adjustable_mel.py:34-37).  Round 6: one arena per
  // geometry, built once and uploaded ASYNCHRONOUSLY from a pinned host image that lives as long as the entry (rounds 1-5 kept ONE arena and
  // synchronised the stream whenever the key shift changed).  LRU of kMelTables: the reference's +-12-semitone augmentation set is 25.
  struct MelTable { int n_fft = 0, win = 0; fdx::DevBuf dev; void* host = nullptr; size_t floats = 0, dft_floats = 0; uint64_t last_use = 0; };
  static constexpr int kMelTables = 32;
  std::vector<MelTable*> mel_tables;
  uint64_t mel_clock = 0;
  long mel_builds = 0, mel_syncs = 0;    // fdx_mel_stats: tables built; stream synchronisations the mel path performed (0 unless the LRU evicts)

  // ---- debug / profiling
  // small per-call scratch of the product paths (spec stats, rand_ini copies): eager launches only, never inside a recorded sampler body --
  // so allocating it (fdx_denorm_spec right behind a handle's first sampler run) must not invalidate that run's recording (round 6: every handle
  // recorded its first shape twice)
  fdx::DevBuf scratch_a{nullptr, 0, false}, scratch_b{nullptr, 0, false};
  fdx::DevBuf dbg_w, dbg_x, dbg_b;    // fdx_debug_conv1d only
  fdx::ProfEvents prof;
  uint64_t alloc_gen = 0;             // see fdx::tl_alloc_gen
};

namespace fdx {
struct GenScope {   // first statement of every entry point that may (re)allocate context buffers
  uint64_t* prev;
  explicit GenScope(fdx_ctx* h) : prev(tl_alloc_gen) { tl_alloc_gen = h ? &h->alloc_gen : nullptr; }
  ~GenScope() { tl_alloc_gen = prev; }
};
}  // namespace fdx

void fdx_rg_free(void* p);   // refinegan.hip
void fdx_mel_free_tables(fdx_ctx* h);   // mel.hip
void fdx_cn_free(void* p);   // convnext.hip
bool fdx_cn_has_attention(fdx_ctx* h);   // cross_attention > 0: an exact-ragged run needs the item layout
// convnext.hip: the two hooks fdx_sampler_run needs (same contracts as wn_embed / wn_forward_core in wavenet.hip)
int fdx_cn_embed(fdx_ctx* h, const float* t_dev, int n, hipStream_t s);
// `fuse` (optional): the UniPC corrector (+ next predictor) applied in the last projection's epilogue instead of storing eps (EpiUniPC, convgemm.hip.h)
int fdx_cn_forward_core(fdx_ctx* h, const float* xin, int col0, int sb_bs, const uint8_t* mask, float* eps_out, long o_bs, int ldo,
                        hipStream_t s, bool unmasked_cond, const fdx::EpiUniPC* fuse = nullptr);
int fdx_cn_plms_setup(fdx_ctx* h, hipStream_t s);   // PLMS + cond_masks: projections of the unmasked condition
void fdx_td_free(void* p);   // tfdec.hip, same hooks
int fdx_td_embed(fdx_ctx* h, const float* t_dev, int n, hipStream_t s);
int fdx_td_forward_core(fdx_ctx* h, const float* xin, int col0, int sb_bs, const uint8_t* mask, float* eps_out, long o_bs, int ldo,
                        hipStream_t s, bool unmasked_cond, const fdx::EpiUniPC* fuse = nullptr);

namespace fdx {

extern thread_local std::string g_last_error;   // defined in core.hip

inline int fail(fdx_ctx* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  g_last_error = buf;
  return code;
}

#define FDX_HIP(h, expr)                                                                      \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fdx::fail(h, FDX_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

inline hipStream_t as_stream(fdx_stream s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace fdx
