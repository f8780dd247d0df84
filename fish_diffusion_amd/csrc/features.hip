// features.hip -- the condition front end that feeds the sampler: DiffSinger.forward_features for the NaiveProjection
// encoders the SVC configs use (archs/diffsinger/diffsinger.py:57-134, modules/encoders/naive_projection.py:6-60,
// utils/pitch.py:12-22).  One fused launch:
//
//   features[b][t][:] = W_text . contents[b][t][:] + b_text          text_encoder (nn.Linear)            diffsinger.py:83
//                       (+ term_0) (+ term_1) ...                     in the reference's order, one rounding per add:
//        speaker:      table[id[b]][:]  or a float mix [B][E] / [B][T][E]                                  :92-108
//        pitch:        w_p * pitch_to_scale(f0[b][t]) + b_p            Linear(1 -> E) after preprocessing  :110-111
//        pitch_shift:  w_s * shift[b] + b_s   (Linear(1 -> E) on [B,1])                                   :113-119
//        energy:       w_e * energy[b][t] + b_e                                                            :121-127
//
// It is < 0.3 % of the path's FLOPs (0.11 GFLOP per 10 s utterance), so this is a plain LDS-tiled fp32 VALU kernel: what
// matters is that the last torch kernels between the feature extractor and the denoiser are gone, not its roofline.
#include "common.hip.h"

using namespace fdx;

namespace {

constexpr int kTT = 32;   // frames per block
constexpr int kTE = 64;   // output channels per block
constexpr int kTC = 32;   // contraction chunk

struct FeatArgs {
  const float* x; const float* w; const float* bias;   // contents [B][T][Din], W [E][Din], bias [E] or null
  float* out;                                           // [B][T][E], or [B][E][T] when channel_first
  const uint8_t* mask;                                  // [B][T] bytes (1 = padding -> output 0) or null
  int B, T, Din, E, n_terms, act, channel_first;
  int S, x_channel_first;                               // contents frames / layout (see fdx_features_forward_src)
  float x_scale;                                        // (float)S / T, the scale F.interpolate(mode="nearest") uses
  const long long* gather;                              // phones2mel [B][T] or null
  const uint8_t* gmask;                                 // [B][T]: 1 => gathered text features * 0 (before the terms)
  int neck; const float* neck_w; const float* neck_b;   // use_neck text encoder: Linear(Din, neck) then w = [E][neck]
  fdx_feature_term terms[FDX_MAX_FEATURE_TERMS];
  float term_scale[FDX_MAX_FEATURE_TERMS];              // (float)src_frames / T per term
};

// F.interpolate(mode="nearest") source index (ATen nearest_neighbor_compute_source_index): fp32 product, floor, clamp
__device__ __forceinline__ int nearest_src(int t, float scale, int S) { return min((int)floorf((float)t * scale), S - 1); }

__global__ __launch_bounds__(256) void k_features(FeatArgs a) {
  __shared__ float xs[kTT][kTC + 1];
  __shared__ float ws[kTE][kTC + 1];
  const int b = blockIdx.z, t0 = blockIdx.x * kTT, e0 = blockIdx.y * kTE;
  const int tid = threadIdx.x;
  const int te = tid & 63, tt = tid >> 6;          // thread -> output channel e0+te, frames t0 + tt + 4*i (i < 8)
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  auto src_frame = [&](int t) {
    if (a.gather) return (int)a.gather[(long)b * a.T + t];
    return a.S == a.T ? t : nearest_src(t, a.x_scale, a.S);
  };
  auto x_at = [&](int ts, int c) { return a.x_channel_first ? a.x[((long)b * a.Din + c) * a.S + ts] : a.x[((long)b * a.S + ts) * a.Din + c]; };
  // use_neck: the bottleneck activations of this block's frames first (rounded to fp32 like the reference's first Linear), then
  // the main loop contracts over them instead of over the contents channels
  __shared__ float hs[kTT][FDX_MAX_NECK + 1];
  if (a.neck > 0) {
    for (int idx = tid; idx < kTT * a.neck; idx += 256) {
      const int r = idx / a.neck, n = idx - r * a.neck, t = t0 + r;
      float hv = 0.f;
      if (t < a.T) {
        const int ts = src_frame(t);
        for (int c = 0; c < a.Din; ++c) hv += x_at(ts, c) * a.neck_w[(long)n * a.Din + c];
        if (a.neck_b) hv += a.neck_b[n];
      }
      hs[r][n] = hv;
    }
    __syncthreads();
  }
  const int Dk = a.neck > 0 ? a.neck : a.Din;           // contraction length of the main loop (= row length of a.w)
  for (int c0 = 0; c0 < Dk; c0 += kTC) {
    for (int i = tid; i < kTT * kTC; i += 256) {
      const int r = i / kTC, c = i - r * kTC;
      const int t = t0 + r;
      float xv = 0.f;
      if (t < a.T && c0 + c < Dk) xv = a.neck > 0 ? hs[r][c0 + c] : x_at(src_frame(t), c0 + c);
      xs[r][c] = xv;
    }
    for (int i = tid; i < kTE * kTC; i += 256) {
      const int r = i / kTC, c = i - r * kTC;
      const int e = e0 + r;
      ws[r][c] = (e < a.E && c0 + c < Dk) ? a.w[(long)e * Dk + c0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < kTC; ++c) {
      const float wv = ws[te][c];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += xs[tt + 4 * i][c] * wv;
    }
    __syncthreads();
  }
  const int e = e0 + te;
  if (e >= a.E) return;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int t = t0 + tt + 4 * i;
    if (t >= a.T) continue;
    float v = acc[i];
    if (a.bias) v += a.bias[e];
    if (a.gmask && a.gmask[(long)b * a.T + t]) v = 0.f;       // gathered features * (1 - mel_mask), diffsinger.py:88-90
    for (int k = 0; k < a.n_terms; ++k) {
      const fdx_feature_term& m = a.terms[k];
      const int Sk = m.src_frames > 0 ? m.src_frames : a.T;
      const int tk = Sk == a.T ? t : nearest_src(t, a.term_scale[k], Sk);
      if (m.kind == FDX_TERM_VECTOR) {            // [B][E] or [B][T][E] float embedding (speaker mix)
        const float* p = static_cast<const float*>(m.values);
        v += m.per_frame ? p[((long)b * Sk + tk) * a.E + e] : p[(long)b * a.E + e];
      } else if (m.kind == FDX_TERM_EMBEDDING) {  // nn.Embedding lookup, ids [B] int64
        const long id = static_cast<const long long*>(m.values)[b];
        v += m.w[id * a.E + e];
      } else {                                    // Linear(1 -> E) on a scalar channel
        const float* p = static_cast<const float*>(m.values);
        float s = m.per_frame ? p[(long)b * Sk + tk] : p[b];
        if (m.preproc == FDX_PRE_PITCH_TO_SCALE) {  // utils/pitch.py:12-22
          s = (s - m.p0) / (m.p1 - m.p0);
          s = s < 0.f ? 0.f : s;
          s = s > 1.f ? 1.f : s;
        }
        float y;
        if (m.neck > 0) {                         // Linear(1, neck) then Linear(neck, E)
          y = 0.f;
          for (int n = 0; n < m.neck; ++n) {
            float hn = m.neck_w[n] * s;
            if (m.neck_b) hn += m.neck_b[n];
            y += m.w[(long)e * m.neck + n] * hn;
          }
        } else {
          y = m.w[e] * s;
        }
        if (m.b) y += m.b[e];
        v += y;
      }
    }
    if (a.act == FDX_ACT_SILU) v = v / (1.f + expf(-v));      // nn.SiLU
    if (a.mask && a.mask[(long)b * a.T + t]) v = 0.f;
    if (a.channel_first) a.out[((long)b * a.E + e) * a.T + t] = v;
    else a.out[((long)b * a.T + t) * a.E + e] = v;
  }
}

__global__ void k_repeat_expand(float* __restrict__ dst, const float* __restrict__ src, int S, int T, float scale) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const long r = blockIdx.y;
  dst[r * T + t] = src[r * S + nearest_src(t, scale, S)];
}

}  // namespace

extern "C" int fdx_repeat_expand(fdx_handle h, const float* src, long rows, int S, int T, float* dst, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (!src || !dst || rows <= 0 || S <= 0 || T <= 0 || rows > 65535) return fail(h, FDX_E_ARG, "fdx_repeat_expand: bad arguments");
  FDX_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(k_repeat_expand, dim3((T + 255) / 256, (unsigned)rows), dim3(256), 0, as_stream(st), dst, src, S, T, (float)S / (float)T);
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}

extern "C" int fdx_features_forward(fdx_handle h, const float* contents, int B, int T, int Din, int E, const float* w_text,
                                    const float* b_text, const fdx_feature_term* terms, int n_terms, float* features,
                                    fdx_stream st) {
  GenScope gen_scope(h);
  return fdx_features_forward_ex(h, contents, B, T, Din, E, w_text, b_text, terms, n_terms, FDX_ACT_NONE, nullptr, 0, features, st);
}

extern "C" int fdx_features_forward_ex(fdx_handle h, const float* contents, int B, int T, int Din, int E, const float* w_text,
                                       const float* b_text, const fdx_feature_term* terms, int n_terms, int act,
                                       const uint8_t* mask, int channel_first, float* features, fdx_stream st) {
  GenScope gen_scope(h);
  return fdx_features_forward_src(h, contents, B, T, 0, T, Din, E, w_text, b_text, terms, n_terms, act, mask, channel_first, features, st);
}

extern "C" int fdx_features_forward_src(fdx_handle h, const float* contents, int B, int S, int contents_channel_first, int T,
                                        int Din, int E, const float* w_text, const float* b_text, const fdx_feature_term* terms,
                                        int n_terms, int act, const uint8_t* mask, int channel_first, float* features,
                                        fdx_stream st) {
  GenScope gen_scope(h);
  return fdx_features_forward_svs(h, contents, B, S, contents_channel_first, T, Din, E, w_text, b_text, 0, nullptr, nullptr, nullptr, nullptr,
                                  terms, n_terms, act, mask, channel_first, features, st);
}

extern "C" int fdx_features_forward_svs(fdx_handle h, const float* contents, int B, int S, int contents_channel_first, int T, int Din,
                                        int E, const float* w_text, const float* b_text, int neck, const float* neck_w,
                                        const float* neck_b, const long long* phones2mel, const uint8_t* gather_mask,
                                        const fdx_feature_term* terms, int n_terms, int act, const uint8_t* mask, int channel_first,
                                        float* features, fdx_stream st) {
  GenScope gen_scope(h);
  if (!h) return FDX_E_ARG;
  if (S <= 0) return fail(h, FDX_E_ARG, "fdx_features_forward: contents has no frames");
  if (act != FDX_ACT_NONE && act != FDX_ACT_SILU) return fail(h, FDX_E_ARG, "fdx_features_forward_ex: unknown activation %d", act);
  if (!contents || !w_text || !features || B <= 0 || T <= 0 || Din <= 0 || E <= 0)
    return fail(h, FDX_E_ARG, "fdx_features_forward: bad arguments");
  if (n_terms < 0 || n_terms > FDX_MAX_FEATURE_TERMS || (n_terms && !terms))
    return fail(h, FDX_E_ARG, "fdx_features_forward: at most %d additive terms", FDX_MAX_FEATURE_TERMS);
  if (neck < 0 || neck > FDX_MAX_NECK || (neck > 0 && !neck_w))
    return fail(h, FDX_E_ARG, "fdx_features_forward_svs: neck size %d (at most %d, with its weights)", neck, FDX_MAX_NECK);
  FeatArgs a{};
  a.x = contents; a.w = w_text; a.bias = b_text; a.out = features;
  a.B = B; a.T = T; a.Din = Din; a.E = E; a.n_terms = n_terms;
  a.act = act; a.mask = mask; a.channel_first = channel_first;
  a.S = S; a.x_channel_first = contents_channel_first; a.x_scale = (float)S / (float)T;
  a.gather = phones2mel; a.gmask = gather_mask; a.neck = neck; a.neck_w = neck_w; a.neck_b = neck_b;
  for (int k = 0; k < n_terms; ++k) {
    const fdx_feature_term& m = terms[k];
    if (m.kind < FDX_TERM_VECTOR || m.kind > FDX_TERM_SCALAR_LINEAR || !m.values)
      return fail(h, FDX_E_ARG, "fdx_features_forward: term %d is malformed", k);
    if (m.kind != FDX_TERM_VECTOR && !m.w) return fail(h, FDX_E_ARG, "fdx_features_forward: term %d has no weights", k);
    if (m.kind == FDX_TERM_SCALAR_LINEAR && m.preproc == FDX_PRE_PITCH_TO_SCALE && m.p1 == m.p0)
      return fail(h, FDX_E_ARG, "fdx_features_forward: term %d: f0_max == f0_min", k);
    if (m.src_frames < 0) return fail(h, FDX_E_ARG, "fdx_features_forward: term %d: negative src_frames", k);
    if (m.neck < 0 || m.neck > FDX_MAX_NECK || (m.neck > 0 && (m.kind != FDX_TERM_SCALAR_LINEAR || !m.neck_w)))
      return fail(h, FDX_E_ARG, "fdx_features_forward: term %d: bad neck", k);
    a.terms[k] = m;
    a.term_scale[k] = (float)(m.src_frames > 0 ? m.src_frames : T) / (float)T;
  }
  FDX_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(k_features, dim3((T + kTT - 1) / kTT, (E + kTE - 1) / kTE, B), dim3(256), 0, as_stream(st), a);
  FDX_HIP(h, hipGetLastError());
  return FDX_OK;
}
