// refinegan_kernels.hip.h -- the non-GEMM kernels of the RefineGAN generator
// (fish_diffusion/modules/vocoders/refinegan/generator.py): comb-tooth template (:174-194), 1-channel convs
// (template_conv :333-341, source_conv :383-389), linear down/up-sampling (nn.Upsample(mode="linear"), :351,394),
// AdaIN noise injection (:104-107).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nsf_kernels.hip.h"

namespace fdx {

// ------------------------------------------------------------------------------------------------ comb-tooth template
// x = cumsum(f0 / sr); x -= round(x); comb = sinc(sr * x / (f0 + 1e-3)) * amp; out = comb * uv + noise_amp * noise.
// torch.cumsum on CPU accumulates in double and rounds every output to fp32: same blocked fp64 scan as the NSF source.
static __global__ __launch_bounds__(kScanThreads) void k_comb_partial(double* __restrict__ partial, const float* __restrict__ f0up,
                                                               int L, int n_chunks, float sr) {
  __shared__ double lds[4];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const float* f = f0up + (long)b * L;
  const int n0 = chunk * kScanChunk + threadIdx.x * kScanPer;
  double s = 0;
  for (int k = 0; k < kScanPer; ++k) {
    const int n = n0 + k;
    if (n < L) s += (double)(f[n] / sr);
  }
  double total;
  (void)block_exclusive_scan(s, lds, total);
  if (threadIdx.x == 0) partial[(long)b * n_chunks + chunk] = total;
}

static __global__ __launch_bounds__(kScanThreads) void k_comb_final(float* __restrict__ out, long out_bs, const double* __restrict__ offsets,
                                                             const float* __restrict__ f0up, const float* __restrict__ noise,
                                                             int L, int n_chunks, float sr, float wave_amp, float noise_std) {
  __shared__ double lds[4];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const float* f = f0up + (long)b * L;
  const int n0 = chunk * kScanChunk + threadIdx.x * kScanPer;
  float v[kScanPer];
  double s = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    const int n = n0 + k;
    v[k] = n < L ? f[n] / sr : 0.f;
    s += (double)v[k];
  }
  double total;
  double run = block_exclusive_scan(s, lds, total) + offsets[(long)b * n_chunks + chunk];
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    const int n = n0 + k;
    run += (double)v[k];
    if (n < L) {
      const float f0v = f[n];
      float x = (float)run;
      x = x - rintf(x);                                         // torch.round: half to even
      const float y = sr * x / (f0v + 1e-3f);
      float sc = 1.f;
      if (y != 0.f) { const float p = 3.14159265358979323846f * y; sc = sinf(p) / p; }   // torch.sinc
      const float comb = sc * wave_amp;
      const float uv = f0v > 0.f ? 1.f : 0.f;
      const float namp = uv * noise_std + (1.f - uv) * wave_amp / 3.f;
      out[b * out_bs + n] = comb * uv + namp * noise[(long)b * L + n];
    }
  }
}

// ------------------------------------------------------------------------------------------------ 1-input-channel convs
// y[b][c][n] (op)= bias[c] + sum_k w[c][k] * src[b][n*stride + k - pad].  MODE 0: y = lrelu(v, slope) (template_conv followed
// by the first leaky_relu, generator.py:449,454); MODE 1: y += v (source_conv, :470-471).  Register-window form (see
// k_noise_conv_add_win): thread n holds its K source samples, channel weights are wave-uniform scalars.
template <int K, int MODE>
static __global__ __launch_bounds__(256) void k_conv1ch(float* __restrict__ y, long y_bs, int ldy, const float* __restrict__ src,
                                                        long src_bs, const float* __restrict__ w, const float* __restrict__ bias,
                                                        int C, int CG, int Lout, int stride, int pad, float slope) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * CG;
  float h[K];
  const float* hp = src + b * src_bs + (long)min(n, Lout - 1) * stride - pad;
#pragma unroll
  for (int k = 0; k < K; ++k) h[k] = hp[k];
  if (n >= Lout) return;
  const int c1 = min(C, c0 + CG);
  for (int c = c0; c < c1; ++c) {
    const float* wp = w + (long)c * K;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) acc += wp[k] * h[k];
    const long o = b * y_bs + (long)c * ldy + n;
    if (MODE == 0) {
      const float v = acc + bias[c];
      y[o] = v > 0.f ? v : v * slope;
    } else {
      y[o] = y[o] + (acc + bias[c]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ linear resampling
// nn.Upsample(scale_factor=1/rate, mode="linear") (align_corners=False): src = rate*(n + 0.5) - 0.5 = rate*n + (rate-1)/2
// -> out[n] = 0.5 * in[rate*n + rate/2 - 1] + 0.5 * in[rate*n + rate/2] for even rates (all shipped configs); general form kept.
static __global__ void k_resample_down(float* __restrict__ out, long o_bs, int ldo, const float* __restrict__ in, long i_bs, int ldi,
                                       int C, int Lin, int Lout, int rate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Lout) return;
  const int b = blockIdx.y / C, c = blockIdx.y - b * C;
  float src = (float)rate * ((float)n + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  const int i0 = (int)src;
  const int i1 = i0 + (i0 < Lin - 1 ? 1 : 0);
  const float l1 = src - (float)i0, l0 = 1.f - l1;
  const float* p = in + b * i_bs + (long)c * ldi;
  out[b * o_bs + (long)c * ldo + n] = l0 * p[i0] + l1 * p[i1];
}

// x = leaky_relu(x, slope); x = nn.Upsample(scale_factor=rate, mode="linear")(x)   (generator.py:466-467)
static __global__ void k_lrelu_resample_up(float* __restrict__ out, long o_bs, int ldo, const float* __restrict__ in, long i_bs, int ldi,
                                           int C, int Lin, int Lout, int rate, float slope) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Lout) return;
  const int b = blockIdx.y / C, c = blockIdx.y - b * C;
  const float scale = 1.f / (float)rate;
  float src = scale * ((float)n + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  const int i0 = (int)src;
  const int i1 = i0 + (i0 < Lin - 1 ? 1 : 0);
  const float l1 = src - (float)i0, l0 = 1.f - l1;
  const float* p = in + b * i_bs + (long)c * ldi;
  float a0 = p[i0], a1 = p[i1];
  a0 = a0 > 0.f ? a0 : a0 * slope;
  a1 = a1 > 0.f ? a1 : a1 * slope;
  out[b * o_bs + (long)c * ldo + n] = l0 * a0 + l1 * a1;
}

// ------------------------------------------------------------------------------------------------ AdaIN
// y = leaky_relu(x + noise * w[c], slope)  (generator.py:104-107).  mode 0: out = y; 1: out += y; 2: out = (out + y) / div
// (the mean over the ParallelResBlock branches, :150-152: torch.mean(torch.stack(results)) = ((r0 + r1) + r2) / 3).
static __global__ void k_adain(float* __restrict__ out, const float* __restrict__ x, long bs, int ld, const float* __restrict__ noise,
                               const float* __restrict__ w, int C, int L, float slope, int mode, float div) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= L) return;
  const int b = blockIdx.y / C, c = blockIdx.y - b * C;
  const long o = b * bs + (long)c * ld + n;
  float v = x[o] + noise[((long)b * C + c) * L + n] * w[c];
  v = v > 0.f ? v : v * slope;
  if (mode == 1) v = out[o] + v;
  else if (mode == 2) v = (out[o] + v) / div;
  out[o] = v;
}

// The same with the noise drawn IN the kernel (device-Philox mode, no injected tensors): round 5 filled a scratch buffer with k_randn and read it back
// here -- 96 extra launches and a third of this kernel's traffic per generator pass (k_randn + k_adain = 10 % of the HiFiSinger line).  A thread owns
// four consecutive frames of one (item, channel) row = exactly one Philox counter of k_randn's stream for this draw (element e = row * L + n, counter
// offset + e / 4; L is a multiple of 4: every stage length is T x a product of the upsampling rates), so the values are k_randn's, bit for bit
// (test: the run equals the one with fdx_randn's tensors injected).
static __global__ void k_adain_rng(float* __restrict__ out, const float* __restrict__ x, long bs, int ld, uint64_t seed, uint64_t offset,
                                   const float* __restrict__ w, int C, int L, float slope, int mode, float div) {
  const int n0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (n0 >= L) return;
  const int b = blockIdx.y / C, c = blockIdx.y - b * C;
  const long o = b * bs + (long)c * ld + n0;
  const uint64_t ctr = offset + (((uint64_t)b * C + c) * (uint64_t)L + n0) / 4;
  uint32_t k[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x66697368u, 0x64782121u};
  philox4x32_10(k, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float r0 = sqrtf(-2.f * logf(u01(k[0]))), r1 = sqrtf(-2.f * logf(u01(k[2])));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u01(k[1]), &s0, &c0);
  sincosf(6.283185307179586f * u01(k[3]), &s1, &c1);
  const float nz[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
  const float4 xv = *reinterpret_cast<const float4*>(x + o);
  float4 ov = mode ? *reinterpret_cast<const float4*>(out + o) : float4{0.f, 0.f, 0.f, 0.f};
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
  float os[4] = {ov.x, ov.y, ov.z, ov.w};
  const float wc = w[c];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = xs[j] + nz[j] * wc;
    v = v > 0.f ? v : v * slope;
    if (mode == 1) v = os[j] + v;
    else if (mode == 2) v = (os[j] + v) / div;
    os[j] = v;
  }
  *reinterpret_cast<float4*>(out + o) = float4{os[0], os[1], os[2], os[3]};
}

}  // namespace fdx
