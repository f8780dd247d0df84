// nsf_kernels.hip.h -- the non-GEMM kernels of the NSF-HiFiGAN generator: harmonic source (f0 upsampling,
// wrap-safe phase scans, sine + noise mix, 9->1 merge), the strided 1-channel noise convs and conv_post.
// Reference: fish_diffusion/modules/vocoders/nsf_hifigan/models.py:195-294 (SineGen), :337-350, :381-393, :434-436.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdx {

// ------------------------------------------------------------------------------------------------ f0 upsampling
// F.interpolate(f0[B,1,T], size=L, mode="linear", align_corners=False)  (models.py:411-413)
static __global__ void k_f0_upsample(float* __restrict__ out, const float* __restrict__ f0, int T, int L) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= L) return;
  const int b = blockIdx.y;
  const float scale = (float)T / (float)L;
  float src = scale * ((float)n + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  const int i0 = (int)src;
  const int i1 = i0 + (i0 < T - 1 ? 1 : 0);
  const float l1 = src - (float)i0, l0 = 1.f - l1;
  const float* f = f0 + (long)b * T;
  out[(long)b * L + n] = l0 * f[i0] + l1 * f[i1];
}

// rad value of harmonic h (0-based) at sample n: ((f0 * (h+1)) / sr) % 1, + rand_ini at n == 0  (models.py:203,214)
__device__ __forceinline__ float rad_of(float f0v, int h, float sr, int n, float rini) {
  float r = fmodf((f0v * (float)(h + 1)) / sr, 1.f);
  if (n == 0) r = r + rini;
  return r;
}

// ------------------------------------------------------------------------------------------------ blocked scans
// torch.cumsum on CPU accumulates fp32 inputs in double and rounds every output to fp32 -- we do the same,
// as a 3-kernel blocked scan: chunk sums -> exclusive scan of chunk sums -> in-chunk scan.
constexpr int kScanThreads = 256;
constexpr int kScanPer = 4;
constexpr int kScanChunk = kScanThreads * kScanPer;   // 1024 samples: 431 chunks per 10 s utterance at 44.1 kHz (4096-sample chunks left 148 of 256 CUs idle at batch 1)

// value fed to the scan of pass P at (b, h, n)
template <int PASS>
__device__ __forceinline__ float scan_input(const float* f0up, const float* tmp, long row_off, int h, int n, float sr,
                                            float rini) {
  const float r = rad_of(f0up[n], h, sr, n, rini);
  if (PASS == 1) return r;
  // pass 2: rad + cumsum_shift, shift[n] = -1 where tmp[n] - tmp[n-1] < 0 (n >= 1)   (models.py:224-231)
  float sh = 0.f;
  if (n > 0 && (tmp[row_off + n] - tmp[row_off + n - 1]) < 0.f) sh = -1.f;
  return r + sh;
}

__device__ __forceinline__ double block_exclusive_scan(double v, double* lds, double& total) {
  // wave-level inclusive scan with shuffles, then across the 4 waves through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    double o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  double base = 0;
  for (int w = 0; w < wave; ++w) base += lds[w];
  total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + inc - v;
}

// partial[b][h][chunk] = sum of the pass's inputs over the chunk
template <int PASS>
static __global__ __launch_bounds__(kScanThreads) void k_scan_partial(double* __restrict__ partial, const float* __restrict__ f0up,
                                                               const float* __restrict__ tmp, const float* __restrict__ rand_ini,
                                                               int L, int H, int n_chunks, float sr) {
  __shared__ double lds[4];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const float* f = f0up + (long)b * L;
  const long row_off = ((long)b * H + h) * L;
  const float rini = rand_ini[b * H + h];
  const int n0 = chunk * kScanChunk + threadIdx.x * kScanPer;
  double s = 0;
  for (int k = 0; k < kScanPer; ++k) {
    const int n = n0 + k;
    if (n < L) s += (double)scan_input<PASS>(f, tmp, row_off, h, n, sr, rini);
  }
  double total;
  (void)block_exclusive_scan(s, lds, total);
  if (threadIdx.x == 0) partial[((long)b * H + h) * n_chunks + chunk] = total;
}

// in-place exclusive scan of each row of `partial` ([rows][n_chunks]); one wave per row (n_chunks ~ 100).  The adds stay one serial
// chain in chunk order (bit-identical to a single thread walking the row); what the wave buys is the loads and stores: 256 chunks
// at a time go through LDS with every lane moving four, instead of one thread waiting for ~100 dependent-looking global loads (16 us).
static __global__ __launch_bounds__(64) void k_scan_offsets(double* __restrict__ partial, int rows, int n_chunks) {
  __shared__ double buf[256], pre[256];            // inputs / exclusive prefixes (two arrays: the serial walk's reads do not wait on its writes)
  const int r = blockIdx.x, lane = threadIdx.x;
  if (r >= rows) return;
  double* p = partial + (long)r * n_chunks;
  double acc = 0;                                  // (lane 0's; carried across slabs)
  for (int c0 = 0; c0 < n_chunks; c0 += 256) {
    const int n = min(256, n_chunks - c0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < n) buf[lane + 64 * j] = p[c0 + lane + 64 * j];
    __syncthreads();
    if (lane == 0) {
#pragma unroll 8
      for (int c = 0; c < n; ++c) {
        pre[c] = acc;
        acc += buf[c];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < n) p[c0 + lane + 64 * j] = pre[lane + 64 * j];
    __syncthreads();
  }
}

// pass 1: tmp[b][h][n] = cumsum(rad)[n] % 1
static __global__ __launch_bounds__(kScanThreads) void k_scan_tmp(float* __restrict__ tmp, const double* __restrict__ offsets,
                                                           const float* __restrict__ f0up, const float* __restrict__ rand_ini,
                                                           int L, int H, int n_chunks, float sr) {
  __shared__ double lds[4];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const float* f = f0up + (long)b * L;
  const long row_off = ((long)b * H + h) * L;
  const float rini = rand_ini[b * H + h];
  const int n0 = chunk * kScanChunk + threadIdx.x * kScanPer;
  float v[kScanPer];
  double s = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    const int n = n0 + k;
    v[k] = n < L ? scan_input<1>(f, nullptr, row_off, h, n, sr, rini) : 0.f;
    s += (double)v[k];
  }
  double total;
  double run = block_exclusive_scan(s, lds, total) + offsets[((long)b * H + h) * n_chunks + chunk];
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    const int n = n0 + k;
    run += (double)v[k];
    if (n < L) tmp[row_off + n] = fmodf((float)run, 1.f);
  }
}

// pass 2 + everything after it: sines = sin(cumsum(rad + shift) * 2 * pi); sine_waves = sines * amp * uv + noise_amp *
// noise; har = tanh(linear(sine_waves))  (models.py:229-231,278-293,346).  One block walks all H harmonics of a chunk.
static __global__ __launch_bounds__(kScanThreads) void k_source_final(float* __restrict__ har, long har_bs, const double* __restrict__ offsets,
                                                               const float* __restrict__ f0up, const float* __restrict__ tmp,
                                                               const float* __restrict__ rand_ini, const float* __restrict__ noise,
                                                               const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                                                               int L, int H, int n_chunks, float sr, float sine_amp,
                                                               float noise_std, float nyquist = 0.f) {   // nyquist > 0: harmonics above it are cleared (RefineGAN's SineGen, generator.py:277-278)
  __shared__ double lds[4];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const float* f = f0up + (long)b * L;
  const int n0 = chunk * kScanChunk + threadIdx.x * kScanPer;
  float acc[kScanPer];
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) acc[k] = 0.f;
  for (int h = 0; h < H; ++h) {
    const long row_off = ((long)b * H + h) * L;
    const float rini = rand_ini[b * H + h];
    const float wh = lin_w[h];
    float v[kScanPer];
    double s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
      const int n = n0 + k;
      v[k] = n < L ? scan_input<2>(f, tmp, row_off, h, n, sr, rini) : 0.f;
      s += (double)v[k];
    }
    double total;
    double run = block_exclusive_scan(s, lds, total) + offsets[((long)b * H + h) * n_chunks + chunk];
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
      const int n = n0 + k;
      run += (double)v[k];
      if (n < L) {
        const float f0v = f[n];
        float sine = sinf((float)run * 2.f * 3.14159265358979323846f);
        if (nyquist > 0.f && f0v * (float)(h + 1) > nyquist) sine = 0.f;
        sine = sine * sine_amp;
        const float uv = f0v > 0.f ? 1.f : 0.f;
        const float namp = uv * noise_std + (1.f - uv) * sine_amp / 3.f;
        const float sw = sine * uv + namp * noise[((long)b * L + n) * H + h];
        acc[k] = acc[k] + sw * wh;
      }
    }
  }
  const float bias = lin_b[0];
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    const int n = n0 + k;
    if (n < L) har[b * har_bs + n] = tanhf(acc[k] + bias);
  }
}

// ------------------------------------------------------------------------------------------------ noise convs
// y[b][c][n] += bias[c] + sum_k w[c][k] * har[b][n*stride + k - pad]      (models.py:381-393,422-423)
// har rows are zero-haloed, so out-of-range taps read 0.
static __global__ void k_noise_conv_add(float* __restrict__ y, long y_bs, int ldy, const float* __restrict__ har, long har_bs,
                                 const float* __restrict__ w, const float* __restrict__ bias, int C, int Lout, int K,
                                 int stride, int pad) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Lout) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const float* hp = har + b * har_bs + (long)n * stride - pad;
  const float* wp = w + (long)c * K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += wp[k] * hp[k];
  const long o = b * y_bs + (long)c * ldy + n;
  y[o] = y[o] + (acc + bias[c]);
}

// Register-window variant: thread n keeps its K-sample window of the source in VGPRs (loaded once, 16-byte loads) and
// walks a group of CG channels; the y read-modify-write is coalesced across the wave.  Same summation order as k_noise_conv_add
// (k ascending).  Stage 0 of config_v1 (C=256, K=128, stride 64): 804 us -> tens of us.
// The weights of up to kSlab / K channels at a time are staged in LDS by the whole workgroup and read back as broadcast
// ds_read_b128 (round 3): as wave-uniform scalar loads straight from memory every 16-weight piece was its own dependent
// s_load -> wait -> 16 multiply-adds round with one wave per SIMD and nothing to hide it behind -- 90 us for stage 0 at batch 1
// whatever the channel grouping.
template <int K>
static __global__ __launch_bounds__(256) void k_noise_conv_add_win(float* __restrict__ y, long y_bs, int ldy,
                                                                   const float* __restrict__ har, long har_bs,
                                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                                   int C, int CG, int Lout, int stride, int pad) {
  constexpr int kSlab = 1024;                      // floats of weights per LDS slab
  constexpr int CH = kSlab / K;                    // channels per slab (8 at K = 128)
  __shared__ __attribute__((aligned(16))) float ws[kSlab];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * CG;
  float h[K];
  const float* hp = har + b * har_bs + (long)min(n, Lout - 1) * stride - pad;   // clamp: overhang lanes read valid memory
#pragma unroll
  for (int k = 0; k < K; ++k) h[k] = hp[k];
  const bool live = n < Lout;                      // (no early return: every thread carries weights into LDS and meets the barriers)
  const int c1 = min(C, c0 + CG);
  for (int cb = c0; cb < c1; cb += CH) {
    const int nc = min(CH, c1 - cb);
    __syncthreads();                               // the previous slab has been consumed
    for (int i = threadIdx.x; i < nc * K; i += 256) ws[i] = w[(long)cb * K + i];
    __syncthreads();
    for (int j = 0; j < nc; ++j) {
      const float* wp = ws + j * K;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) acc += wp[k] * h[k];
      if (live) {
        const long o = b * y_bs + (long)(cb + j) * ldy + n;
        y[o] = y[o] + (acc + bias[cb + j]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ conv_post
// wav[b][n] = tanh(bias + sum_c sum_k w[c][k] * lrelu(x[b][c][n + k - 3], 0.01))   (models.py:434-436)
static __global__ void k_conv_post(float* __restrict__ wav, long wav_bs, const float* __restrict__ x, long x_bs, int ldx,
                            const float* __restrict__ w, const float* __restrict__ bias, int C, int L, float slope) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= L) return;
  const int b = blockIdx.y;
  const float* xp = x + b * x_bs + n - 3;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* xr = xp + (long)c * ldx;
    const float* wr = w + c * 7;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      float v = xr[k];
      v = v > 0.f ? v : v * slope;
      acc += wr[k] * v;
    }
  }
  wav[b * wav_bs + n] = tanhf(acc + bias[0]);
}

}  // namespace fdx
