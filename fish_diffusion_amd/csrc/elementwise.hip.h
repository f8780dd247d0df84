// elementwise.hip.h -- the HBM-trivial kernels around the MFMA contractions: layout staging, the diffusion
// step embedding, the sampler update rules and the Philox generator.  All tensors touched here are
// [B][M][T]-shaped (<= a few MB); one thread per element, rows on blockIdx.y so that every access is a
// coalesced 128-byte-aligned row segment (padded rows start kHalo floats in).
//
// The update rules deliberately spell every product/sum as a separate rounding (contract(off)): the
// reference evaluates them as individual torch ops, and FMA contraction would change the last bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdx {

#pragma clang fp contract(off)

constexpr int kEwBlock = 256;
inline dim3 ew_grid(int T, int rows) { return dim3((T + kEwBlock - 1) / kEwBlock, rows); }

// dst[b][c][t] (pitch ldd, item stride d_bs) = scale * src[b][c][t] (pitch lds, item stride s_bs); masked cols -> 0
static __global__ void k_copy_rows(float* __restrict__ dst, long d_bs, int ldd, const float* __restrict__ src, long s_bs,
                            int lds, int C, int T, float scale, const uint8_t* __restrict__ mask) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const int b = blockIdx.y / C, c = blockIdx.y - b * C;
  float v = src[b * s_bs + (long)c * lds + t];
  if (scale != 1.f) v = scale * v;
  if (mask && mask[(long)b * T + t]) v = 0.f;
  dst[b * d_bs + (long)c * ldd + t] = v;
}

// padded -> padded copy with a leaky-relu (slope 1 = plain copy): dst[b][c][t] = lrelu(src[b][c][t])
static __global__ void k_copy_rows_act(float* __restrict__ dst, long d_bs, int ldd, const float* __restrict__ src, long s_bs, int lds,
                                int C, int T, float slope) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const int b = blockIdx.y / C, c = blockIdx.y - b * C;
  const float v = src[b * s_bs + (long)c * lds + t];
  dst[b * d_bs + (long)c * ldd + t] = v > 0.f ? v : v * slope;
}

// wavenet.py:20-27 -- E[k][j] = sin(t_j * f_k) (k < half) | cos(t_j * f_{k-half}); f_k = exp(k * -(ln 1e4/(half-1)))
static __global__ void k_step_embed(float* __restrict__ E, int ldn, const float* __restrict__ t, int n, int dim) {
  const int j = blockIdx.x * kEwBlock + threadIdx.x;
  if (j >= n) return;
  const int k = blockIdx.y, half = dim / 2;
  const float c = -(logf(10000.f) / (float)(half - 1));   // fp32, as the python-float -> fp32 scalar
  const int kk = k < half ? k : k - half;
  const float f = expf((float)kk * c);
  const float a = t[j] * f;
  E[(long)k * ldn + j] = k < half ? sinf(a) : cosf(a);
}

// ---------------------------------------------------------------- UniPC (uni_pc.py:583-701, predict_x0 branch)
// m = (x - sigma*eps) / alpha                                                   uni_pc.py:348
static __global__ void k_x0_pred(float* __restrict__ m, const float* __restrict__ x, const float* __restrict__ eps,
                          long bs, int ld, int M, int T, float sigma, float alpha) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const long o = (blockIdx.y / M) * bs + (long)(blockIdx.y % M) * ld + t;
  m[o] = (x[o] - sigma * eps[o]) / alpha;
}

// x_base = c_x*x - c_m*m0 ; x_t = x_base - aB*(0.5*D1), D1 = (m1-m0)/rk (order 2)        uni_pc.py:664-671
static __global__ void k_unipc_pre(float* __restrict__ xbase, float* __restrict__ xt, const float* __restrict__ x,
                            const float* __restrict__ m0, const float* __restrict__ m1, long bs, int ld, int M, int T,
                            float c_x, float c_m, float aB, float rk, int order) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const long o = (blockIdx.y / M) * bs + (long)(blockIdx.y % M) * ld + t;
  const float m0v = m0[o];
  const float xb = c_x * x[o] - c_m * m0v;
  float r = xb;
  if (order == 2) {
    const float D1 = (m1[o] - m0v) / rk;
    r = xb - aB * (0.5f * D1);
  }
  xbase[o] = xb;
  xt[o] = r;
}

// model_t = (x_t - sigma*eps)/alpha ; x = x_base - aB*(rho0*D1 + rho1*(model_t - m0))       uni_pc.py:673-680
static __global__ void k_unipc_post(float* __restrict__ x, float* __restrict__ mt, const float* __restrict__ xbase,
                             const float* __restrict__ xt, const float* __restrict__ eps, const float* __restrict__ m0,
                             const float* __restrict__ m1, long bs, int ld, int M, int T, float sigma, float alpha,
                             float aB, float rk, int order, float rho0, float rho1) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const long o = (blockIdx.y / M) * bs + (long)(blockIdx.y % M) * ld + t;
  const float m0v = m0[o];
  const float mtv = (xt[o] - sigma * eps[o]) / alpha;
  const float D1t = mtv - m0v;
  float corr;
  if (order == 2) {
    const float D1 = (m1[o] - m0v) / rk;
    corr = rho0 * D1 + rho1 * D1t;
  } else {
    corr = rho1 * D1t;   // python `0 + rho*D1_t`
  }
  mt[o] = mtv;
  x[o] = xbase[o] - aB * corr;
}

// k_unipc_post of step r immediately followed by k_unipc_pre of step r+1 on the same element (one launch, one round trip
// instead of two; the arithmetic and its order are unchanged).  After the corrector the history shifts: m0' = model_t,
// m1' = m0 (uni_pc.py:797-804), so the next predictor's D1 = (m1' - m0') / rk' needs nothing from memory.
static __global__ void k_unipc_post_pre(float* __restrict__ x, float* __restrict__ mt, float* __restrict__ xbase,
                                 float* __restrict__ xt, const float* __restrict__ eps, const float* __restrict__ m0,
                                 const float* __restrict__ m1, long bs, int ld, int M, int T, float sigma, float alpha,
                                 float aB, float rk, int order, float rho0, float rho1, float n_cx, float n_cm, float n_aB,
                                 float n_rk, int n_order) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const long o = (blockIdx.y / M) * bs + (long)(blockIdx.y % M) * ld + t;
  const float m0v = m0[o];
  const float mtv = (xt[o] - sigma * eps[o]) / alpha;
  const float D1t = mtv - m0v;
  float corr;
  if (order == 2) {
    const float D1 = (m1[o] - m0v) / rk;
    corr = rho0 * D1 + rho1 * D1t;
  } else {
    corr = rho1 * D1t;
  }
  const float xn = xbase[o] - aB * corr;
  mt[o] = mtv;
  x[o] = xn;
  // ---- predictor of the next step (k_unipc_pre with x = xn, m0 = mtv, m1 = m0v)
  const float xb = n_cx * xn - n_cm * mtv;
  float r = xb;
  if (n_order == 2) {
    const float D1n = (m0v - mtv) / n_rk;
    r = xb - n_aB * (0.5f * D1n);
  }
  xbase[o] = xb;
  xt[o] = r;
}

// ---------------------------------------------------------------- DDPM ancestral step (noise_predictor.py:73-104)
static __global__ void k_naive_step(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ noise,
                             long n_bs, int n_ld, long bs, int ld, int M, int T, float sr, float srm1, float c1,
                             float c2, float nscale, float clip_min, float clip_max) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const int b = blockIdx.y / M, m = blockIdx.y % M;
  const long o = b * bs + (long)m * ld + t;
  const float xv = x[o];
  float x0 = sr * xv - srm1 * eps[o];
  x0 = fminf(fmaxf(x0, clip_min), clip_max);   // torch.clamp(min=clip_min, max=clip_max), noise_predictor.py:90
  const float mean = c1 * x0 + c2 * xv;
  x[o] = mean + nscale * noise[b * n_bs + (long)m * n_ld + t];
}

// ---------------------------------------------------------------- PLMS (noise_predictor.py:118-148)
// out = x + A*(P*x - Q*e)
static __global__ void k_plms_pred(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ e, long bs,
                            int ld, int M, int T, float A, float P, float Q) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const long o = (blockIdx.y / M) * bs + (long)(blockIdx.y % M) * ld + t;
  const float xv = x[o];
  out[o] = xv + A * (P * xv - Q * e[o]);
}
// stage 0: (e + h1)/2 ; 1: (3e - h1)/2 ; 2: (23e - 16h1 + 5h2)/12 ; 3: (55e - 59h1 + 37h2 - 9h3)/24
static __global__ void k_plms_blend(float* __restrict__ out, const float* __restrict__ e, const float* __restrict__ h1,
                             const float* __restrict__ h2, const float* __restrict__ h3, long bs, int ld, int M, int T,
                             int stage) {
  const int t = blockIdx.x * kEwBlock + threadIdx.x;
  if (t >= T) return;
  const long o = (blockIdx.y / M) * bs + (long)(blockIdx.y % M) * ld + t;
  const float ev = e[o];
  float r;
  if (stage == 0) r = (ev + h1[o]) / 2.f;
  else if (stage == 1) r = (ev * 3.f - h1[o]) / 2.f;
  else if (stage == 2) r = (ev * 23.f - h1[o] * 16.f + h2[o] * 5.f) / 12.f;
  else r = (ev * 55.f - h1[o] * 59.f + h2[o] * 37.f - h3[o] * 9.f) / 24.f;
  out[o] = r;
}

// ---------------------------------------------------------------- shallow-diffusion entry (diffusion.py:223-232)
// out = q_sample(norm_spec(src)) :  v = (src - smin) / (smax - smin) * 2 - 1   (diffusion.py:315-316; the [1,1,n] stats broadcast
// against the LAST axis of [B,M,T], exactly as the reference's expression does), then  a * v + b * noise  (:120-127).
// Every product / quotient / sum is its own rounding (-ffp-contract=off), in the reference's order.
static __global__ void k_q_sample(float* __restrict__ out, const float* __restrict__ src, const float* __restrict__ noise, size_t n,
                                  int T, int normalise, const float* __restrict__ smin, const float* __restrict__ smax, int n_spec,
                                  int do_q, float a, float b) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = src[i];
  if (normalise) {
    const int k = n_spec == 1 ? 0 : (int)(i % (size_t)T);
    const float lo = smin[k], hi = smax[k];
    v = (v - lo) / (hi - lo) * 2.f - 1.f;
  }
  if (do_q) v = a * v + b * noise[i];
  out[i] = v;
}

// ---------------------------------------------------------------- denorm_spec + transpose (diffusion.py:318-319)
// mel[b][t][m] = (x[b][m][t] + 1)/2 * (smax - smin) + smin ; spec arrays indexed by m (or 0 when n_spec == 1)
static __global__ void k_denorm_transpose(float* __restrict__ mel, const float* __restrict__ x, int M, int T,
                                   const float* __restrict__ smin, const float* __restrict__ smax, int n_spec) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, t = t0 + tx;
    if (m < M && t < T) tile[r][tx] = x[((long)b * M + m) * T + t];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, m = m0 + tx;
    if (m < M && t < T) {
      const float lo = smin[n_spec == 1 ? 0 : m], hi = smax[n_spec == 1 ? 0 : m];
      mel[((long)b * T + t) * M + m] = (tile[tx][r] + 1.f) / 2.f * (hi - lo) + lo;
    }
  }
}

// ---------------------------------------------------------------- Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float u01(uint32_t u) { return ((float)(u >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

// 4 normals per thread; element e = 4*i + k
static __global__ void k_randn(float* __restrict__ out, size_t n, uint64_t seed, uint64_t offset) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 >= n) return;
  const uint64_t ctr = offset + i;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x66697368u, 0x64782121u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  float r0 = sqrtf(-2.f * logf(u01(c[0]))), r1 = sqrtf(-2.f * logf(u01(c[2])));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u01(c[1]), &s0, &c0);
  sincosf(6.283185307179586f * u01(c[3]), &s1, &c1);
  const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
  for (int k = 0; k < 4; ++k)
    if (i * 4 + k < n) out[i * 4 + k] = v[k];
}
static __global__ void k_rand_uniform(float* __restrict__ out, size_t n, uint64_t seed, uint64_t offset) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 >= n) return;
  const uint64_t ctr = offset + i;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x756e6966u, 0x64782121u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  for (int k = 0; k < 4; ++k)
    if (i * 4 + k < n) out[i * 4 + k] = (float)(c[k] >> 8) * (1.0f / 16777216.0f);   // [0,1)
}

}  // namespace fdx
