/*
 * fishdx.h -- C ABI of libfishdx.so: the MI355X (gfx950) native hot path of fish-diffusion's
 * SVC/SVS inference: WaveNet-residual denoiser, noise-schedule sampler loop, STFT/mel front end
 * and NSF-HiFiGAN vocoder.
 *
 * Every entry point replaces one Python-level interface of the reference (cited as file:line,
 * relative to the fish-diffusion v2.2.0 tree).  The reference has no native code, so what a
 * maintainer binds is a ctypes stub inside the registered nn.Module -- see INTEGRATION.md.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  "dev" pointers are HIP device memory owned by the
 *     caller (torch tensors' data_ptr()); "host" pointers are ordinary host memory.
 *   - every function returns 0 on success, <0 on error; fdx_last_error() gives the message.
 *     No exception crosses the ABI; the Python wrappers map codes to the reference's exception types.
 *   - kernels are enqueued on the caller's stream (hipStream_t passed as void*); the library never
 *     synchronises the device inside *_forward / *_run.  Buffers grow only in *_prepare / first call
 *     with a new geometry.
 *   - all arithmetic is fp32 (f32-in/f32-acc MFMA), activations are [B][C][T] with T contiguous.
 *   - multi-GPU: the data path has no collective (utterances shard, nothing is exchanged).  The packed arenas
 *     (fdx_*_pack) are plain byte blobs; rank 0 packs, every rank calls fdx_*_attach.  Shipping them is the host's business:
 *     `torch.distributed.broadcast` (RCCL) in fish_diffusion_amd/dist.py, or -- for a host that is not PyTorch --
 *     fdx_bcast_arena below, ONE ncclBroadcast over a communicator the host created.
 *   - a handle is NOT re-entrant: one in-flight call per handle (the Python wrapper holds a lock;
 *     the reference's flask_api.py:86 can call forward from several threads).
 *   - a handle's workspaces and cached tables are reused from call to call and are ordered by the STREAM the calls are enqueued
 *     on: consecutive calls on one handle that use different streams must be ordered by the caller (event / stream wait), as
 *     for any buffer shared between streams.  Independent streams want independent handles (pipeline.py's vocoder lanes).
 */
#ifndef FISHDX_H_
#define FISHDX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fdx_ctx* fdx_handle;
typedef void* fdx_stream; /* hipStream_t */

enum {
  FDX_OK = 0,
  FDX_E_ARG = -1,     /* bad argument / geometry (maps to ValueError / AssertionError) */
  FDX_E_STATE = -2,   /* call order (weights not attached, prepare not called)  (RuntimeError) */
  FDX_E_HIP = -3,     /* HIP runtime error (RuntimeError) */
  FDX_E_NOIMPL = -4,  /* unsupported variant (NotImplementedError) */
  FDX_E_NOMEM = -5
};

int fdx_version(void);
/* 1 if a HIP device is visible to this process, else 0 (never fails). */
int fdx_device_available(void);
int fdx_create(int device, fdx_handle* out);
int fdx_destroy(fdx_handle h);
/* h may be NULL: returns the last error of a failed fdx_create / pure-host call on this thread. */
const char* fdx_last_error(fdx_handle h);

/* ------------------------------------------------------------------------------------------------
 * WaveNet denoiser  -- replaces fish_diffusion/modules/wavenet.py:151-236 (WaveNet.__init__/forward),
 * registered as DENOISERS "WaveNetDenoiser" at archs/diffsinger/diffusions/builder.py:10.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int mel_channels;      /* 128 */
  int d_encoder;         /* 256 */
  int residual_channels; /* 512; multiple of 32 */
  int residual_layers;   /* 20 */
  int dilation_cycle;    /* 4, or 0 for "None" (all dilations 1) -- wavenet.py:181 */
  int use_linear_bias;   /* wavenet.py:161 */
} fdx_wavenet_desc;

/* Canonical tensor order expected by fdx_wavenet_pack = the reference's state_dict order
 * (wavenet.py:168-191): input_projection.conv.{weight,bias}, mlp.0.linear.{weight[,bias]},
 * mlp.2.linear.{weight[,bias]}, then per layer conv_layer.conv.{weight,bias},
 * diffusion_projection.linear.{weight[,bias]}, conditioner_projection.conv.{weight,bias},
 * output_projection.conv.{weight,bias}; then skip_projection.conv.{weight,bias},
 * output_projection.conv.{weight,bias}.  Conv1d weights [Cout][Cin][k], Linear [out][in]. */
int fdx_wavenet_num_weights(const fdx_wavenet_desc* d);
int fdx_wavenet_packed_bytes(const fdx_wavenet_desc* d, size_t* bytes);
/* Pure host: repack reference-layout fp32 tensors into the MFMA-fragment-ordered arena. */
int fdx_wavenet_pack(const fdx_wavenet_desc* d, const float* const* host_weights, int n_weights,
                     void* host_packed, size_t bytes);
/* Attach a packed arena that already lives in device memory (owned by the caller; it must outlive
 * the handle or the next attach).  Rank 0 packs + uploads, the other ranks receive the same bytes
 * by one RCCL broadcast and attach them -- see fish_diffusion_amd/dist.py. */
int fdx_wavenet_attach(fdx_handle h, const fdx_wavenet_desc* d, const void* dev_packed, size_t bytes);

/* Opt-in bf16 storage mode (BASELINE configs[4] as SURVEY F4 reads it: "bf16 storage / fp32 accumulate"): the two residual-block
 * GEMMs read bf16 weights and bf16 activation operands through v_mfma_f32_32x32x16_bf16; everything else stays fp32.  Not
 * parity-grade (8 mantissa bits): its error is measured and reported separately, it is never the headline path.
 * pack: same tensor list as fdx_wavenet_pack.  attach: after fdx_wavenet_attach; dev_packed = NULL switches back to fp32. */
int fdx_wavenet_bf16_packed_bytes(const fdx_wavenet_desc* d, size_t* bytes);
int fdx_wavenet_bf16_pack(const fdx_wavenet_desc* d, const float* const* host_weights, int n_weights, void* host_packed, size_t bytes);
int fdx_wavenet_bf16_attach(fdx_handle h, const void* dev_packed, size_t bytes);
/* The same bf16 arena, derived on the device from the fp32 arena attached to `h` (bit-identical to fdx_wavenet_bf16_pack of the
 * tensors that arena was packed from): for ranks that received their weights as a broadcast arena and hold no tensors. */
int fdx_wavenet_bf16_from_arena(fdx_handle h, void* dev_out, size_t bytes, fdx_stream s);

/* Opt-in fp16-split mode ("past the fp32 roof"): the same two GEMMs with every operand held as hi + lo fp16 (22 mantissa bits)
 * and each product block formed as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16, fp32 accumulate: fp32-class results (held
 * to the fp32 path's own parity bars by the tests) at up to 5.3x the fp32 MFMA rate.  Taken per call when the launches have
 * enough LDS tiles (batch >= 2 at 10 s), the fp32 kernels otherwise.  The weights are derived on the device from the attached
 * fp32 arena; on = 0 switches back.  Mutually exclusive with the bf16 storage mode.  Never what the default bench line measures. */
int fdx_wavenet_f16s_enable(fdx_handle h, int on);

/* Step-invariant work for one batch of utterances (conditioner slabs for all layers, wavenet.py:108).
 * cond: dev [B][d_encoder][T]; cond_mask: dev [B][T] bytes (1 = masked, wavenet.py:220-221) or NULL. */
int fdx_wavenet_prepare(fdx_handle h, const float* cond, int B, int T, const uint8_t* cond_mask,
                        fdx_stream s);
/* eps = WaveNet(x, t, cond).  x, eps: dev [B][mel_channels][T]; t: dev [n_t] floats with n_t == B or 1
 * (the samplers pass one timestep for the whole batch, noise_predictor.py:12-13);
 * x_mask: dev [B][T] bytes or NULL (wavenet.py:217-218,233-234). */
int fdx_wavenet_forward(fdx_handle h, const float* x, const float* t, int n_t, const uint8_t* x_mask,
                        float* eps, fdx_stream s);

/* ------------------------------------------------------------------------------------------------
 * ConvNext denoiser -- replaces fish_diffusion/modules/convnext.py:155-262 (class ConvNext, with or
 * without cross-attention; blocks :12-92, :95-152), DENOISERS "ConvNextDenoiser"
 * (archs/diffsinger/diffusions/builder.py:12).  Same call contract as the WaveNet: whichever of
 * fdx_wavenet_prepare / fdx_convnext_prepare ran last selects the denoiser fdx_sampler_run drives.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fdx_convnext_desc {
  int mel_channels;    /* 128; multiple of 8 */
  int dim;             /* 512; multiple of 32, <= 512 */
  int mlp_factor;      /* 4 */
  int condition_dim;   /* 256; multiple of 8 */
  int num_layers;      /* 20 */
  int dilation_cycle;  /* 4 (dilations 2^(i % cycle), convnext.py:200) */
  int cross_attention; /* 0 = off; n > 0 = `cross_every_n_layers` of the reference: a CrossAttentionBlock (convnext.py:95-152,186-193) in
                        * front of every n-th ConvNext block, its tensors in the module's state_dict order (csrc/convnext.hip:47-93) */
} fdx_convnext_desc;
/* Canonical tensor order = the module's state_dict order (convnext.py:170-205): input_projection.{weight,bias},
 * diffusion_embedding.{1,3}.{weight,bias}, conditioner_projection.{0,2}.{weight,bias}, per layer gamma,
 * dwconv.{weight,bias}, norm.{weight,bias}, pwconv1.*, pwconv2.*, diffusion_step_projection.*, condition_projection.*;
 * then output_projection.{0,2}.{weight,bias}. */
int fdx_convnext_num_weights(const fdx_convnext_desc* d);
int fdx_convnext_packed_bytes(const fdx_convnext_desc* d, size_t* bytes);
int fdx_convnext_pack(const fdx_convnext_desc* d, const float* const* host_weights, int n_weights,
                      void* host_packed, size_t bytes);
int fdx_convnext_attach(fdx_handle h, const fdx_convnext_desc* d, const void* dev_packed, size_t bytes);
/* cond: dev [B][condition_dim][T]; cond_mask: dev [B][T] bytes or NULL (convnext.py:247-248,69-70). */
int fdx_convnext_prepare(fdx_handle h, const float* cond, int B, int T, const uint8_t* cond_mask,
                         fdx_stream s);
/* eps = ConvNext(x, t, cond); arguments as fdx_wavenet_forward. */
int fdx_convnext_forward(fdx_handle h, const float* x, const float* t, int n_t, const uint8_t* x_mask,
                         float* eps, fdx_stream s);

/* ------------------------------------------------------------------------------------------------
 * Transformer-decoder denoiser -- replaces fish_diffusion/modules/convnext.py:263-379
 * (class TransformerDecoderDenoiser: 1x1-conv projections around num_layers x nn.TransformerDecoderLayer,
 * 8 heads, post-norm, GELU), DENOISERS "TransformerDecoderDenoiser" (diffusions/builder.py:13).
 * Same call contract; the conditioner must have as many frames as the mel (how GaussianDiffusion calls it).
 * ---------------------------------------------------------------------------------------------- */
typedef struct fdx_tfdec_desc {
  int mel_channels;   /* 128; multiple of 8 */
  int dim;            /* 512 (also 256, 128): 8 heads of dim/8 channels */
  int mlp_factor;     /* 4 */
  int condition_dim;  /* 256; multiple of 8 */
  int num_layers;     /* 12 */
  int n_positions;    /* 4096: rows of the positional_embedding buffer (convnext.py:317) */
} fdx_tfdec_desc;
/* Canonical tensor order = the module's state_dict order: position_scale_query, position_scale_key, positional_embedding,
 * input_projection.{0,2}.*, diffusion_embedding.{1,3}.*, condition_projection.{0,2}.*, per layer self_attn.{in_proj_weight,
 * in_proj_bias, out_proj.weight, out_proj.bias}, multihead_attn.(same), linear1.*, linear2.*, norm1.*, norm2.*, norm3.*;
 * output_projection.{0,2}.*. */
int fdx_tfdec_num_weights(const fdx_tfdec_desc* d);
int fdx_tfdec_packed_bytes(const fdx_tfdec_desc* d, size_t* bytes);
int fdx_tfdec_pack(const fdx_tfdec_desc* d, const float* const* host_weights, int n_weights, void* host_packed, size_t bytes);
int fdx_tfdec_attach(fdx_handle h, const fdx_tfdec_desc* d, const void* dev_packed, size_t bytes);
int fdx_tfdec_prepare(fdx_handle h, const float* cond, int B, int T, const uint8_t* cond_mask, fdx_stream s);
int fdx_tfdec_forward(fdx_handle h, const float* x, const float* t, int n_t, const uint8_t* x_mask, float* eps, fdx_stream s);

/* ------------------------------------------------------------------------------------------------
 * Weights over xGMI -- SURVEY 8(b)'s `fdx_bcast_weights`.  The reference's workers each load the checkpoint themselves
 * (tools/preprocessing/extract_features.py:262-322); here rank `root` packs once (fdx_*_pack) and every rank receives the arena:
 * ONE ncclBroadcast of `bytes` bytes, in place, on the caller's stream, over a communicator the HOST created (ncclComm_t passed as
 * void*).  RCCL is bound at call time (dlopen("librccl.so.1")): the library does not link against it.  Every rank then calls
 * fdx_*_attach on its copy.  (The Python host does the same with torch.distributed.broadcast: dist.py::broadcast_model_weights.)
 * FDX_E_NOIMPL if librccl cannot be loaded, FDX_E_HIP with RCCL's message if the collective fails.
 * ---------------------------------------------------------------------------------------------- */
int fdx_bcast_arena(void* dev_arena, size_t bytes, void* rccl_comm, int root, fdx_stream s);

/* ------------------------------------------------------------------------------------------------
 * Sampler loop -- replaces GaussianDiffusion.forward's loop, diffusion.py:234-311, and the three
 * predictors (noise_predictor.py:19-222, uni_pc.py:583-818).  The per-step scalars are computed by
 * the host (fish_diffusion_amd/schedule.py mirrors the reference's fp32 arithmetic) and passed as a
 * table of FDX_ROW floats per row; the tensor math runs on the device with no host round trip.
 * ---------------------------------------------------------------------------------------------- */
enum { FDX_SAMPLER_NAIVE = 0, FDX_SAMPLER_UNIPC = 1, FDX_SAMPLER_PLMS = 2 };
#define FDX_ROW 16
/* Row layouts (unused columns 0):
 *  UNIPC row 0      : {t_input, sigma_t, alpha_t}                                  (initial model call)
 *  UNIPC row r>=1   : {t_input, sigma_t, alpha_t, c_x = sigma_t/sigma_prev, c_m = alpha_t*h_phi_1,
 *                      aB = alpha_t*B_h, rk, order(1|2), use_corrector(0|1), rho_c0, rho_c1}
 *  NAIVE row        : {t, sqrt_recip_ac, sqrt_recipm1_ac, coef1, coef2, noise_scale, clip_min, clip_max}
 *  PLMS  row        : {t, t_prev, A = a_prev - a_t, P, Q}   (noise_predictor.py:118-131)
 */
/* x: dev [B][M][T], in = x_T (initial noise or q_sample'd mel), out = x_0 (normalised mel).
 * step_noise: NAIVE only, dev [n_rows][B][M][T] standard normals, or NULL => device Philox(seed).
 * x_mask as in fdx_wavenet_forward.  fdx_wavenet_prepare must have been called for this batch. */
int fdx_sampler_run(fdx_handle h, int kind, const float* host_table, int n_rows, float* x,
                    const float* step_noise, uint64_t seed, const uint8_t* x_mask, fdx_stream s);
/* The same sampler run in EXACT-MASK mode: frames with x_mask != 0 (dev [B][T] bytes) are treated as NON-EXISTENT -- the dilated convs
 * read zeros there at every layer, exactly like their own zero padding -- instead of the reference's masked semantics, where a masked
 * frame is zeroed at the denoiser's input and output but stays alive in between (wavenet.py:217-221,233-234) and so leaks into the
 * receptive field of its neighbours.  Consequence: any run of >= 16 masked frames isolates what lies on either side of it, bit for
 * bit.  That is what lets a serving loop lay a ragged batch out as ONE row -- items separated by 16-frame holes, no padding to a
 * common length -- and get every item exactly as if it had been run alone (what tools/diffusion/inference.py:336-376 computes
 * one segment at a time); fish_diffusion_amd.GaussianDiffusion(..., lengths=) does that.  x at masked frames is left undefined.
 * WaveNet denoiser.  fp32 storage: "bit for bit" holds for every geometry.  Opt-in fp16-split storage (fdx_wavenet_f16s_enable): the
 * exact-mask epilogues exist for both of its kernel families, but a long ragged row may run the 128-wide LDS tiles (bf16lds.hip.h) where a
 * short item alone runs the 64 x 64 tiles (f16s64.hip.h) -- chosen by tile count, different summation grouping -- so an item agrees with
 * its stand-alone run to fp32 rounding, not bit for bit.  Not available in bf16 storage (FDX_E_NOIMPL). */
int fdx_sampler_run_ragged(fdx_handle h, int kind, const float* host_table, int n_rows, float* x, const float* step_noise,
                           uint64_t seed, const uint8_t* x_mask, fdx_stream s);
/* Exact-mask runs of the OTHER denoisers behind the same contract (modules/convnext.py:211 ConvNext.forward, :325 TransformerDecoderDenoiser.forward;
 * round 5).  The ConvNeXt blocks are a depthwise conv (k = 7, dilation <= 2^(cycle-1): reach 3 * 2^(cycle-1) frames) between per-frame ops, so a
 * hole at least that wide isolates its sides exactly as for the WaveNet and fdx_sampler_run_ragged needs nothing more.  Attention does not stop
 * at holes and the positional table is indexed by the frame's position IN ITS UTTERANCE, so the attention-based denoisers (the transformer;
 * ConvNext with cross-attention) must be told where the items of the ONE row (B == 1) lie: call this BEFORE fdx_*_prepare with host arrays of
 * n_items offsets (multiples of 32, ascending, non-overlapping) and lengths; self- and cross-attention then run per item over its own frames
 * and positions restart at each offset.  n_items == 0 clears the layout (dense batches).  T = length of the row the items lie in.  Every item
 * comes out bit-identical to a batch-1 run of it alone (the key split of an item's attention depends on its own length only).
 * fdx_sampler_run_ragged REFUSES (FDX_E_STATE) to run an attention-based denoiser without a layout: the result would silently not be the
 * per-item one.  A rejected layout (FDX_E_ARG) leaves the handle with NO layout, never half of one.  The offsets / lengths are copied
 * during the call (they travel to the device as kernel arguments): the host arrays may be freed on return, nothing synchronises. */
int fdx_sampler_set_items(fdx_handle h, const int* host_offsets, const int* host_lens, int n_items, int T, fdx_stream s);
/* The start of shallow diffusion, diffusion.py:223-232: out = q_sample(norm_spec(src), t, noise).
 *   normalise != 0: v = (src - spec_min) / (spec_max - spec_min) * 2 - 1 (diffusion.py:315-316).  spec_min/max are host arrays of
 *   n_spec floats; the reference's [1,1,n] buffers broadcast against the LAST axis of the [B,M,T] tensor, so n_spec is 1 or T.
 *   noise != NULL: out = sqrt_ac * v + sqrt_1m_ac * noise (q_sample :120-127 with extract() :34-37 done by the caller: the two
 *   scalars are sqrt_alphas_cumprod[t], sqrt_one_minus_alphas_cumprod[t] for t = timesteps - skip_steps).
 * src, noise, out: dev [B][M][T]; out may alias src. */
int fdx_q_sample(fdx_handle h, const float* src, int B, int M, int T, int normalise, const float* spec_min, const float* spec_max,
                 int n_spec, float sqrt_ac, float sqrt_1m_ac, const float* noise, float* out, fdx_stream s);
/* norm_spec / denorm_spec + the [B,M,T] <-> [B,T,M] transposes, diffusion.py:315-319,217.
 * spec_min/max: host arrays of n_spec (1 or M) floats.  denorm: x [B][M][T] -> mel [B][T][M]. */
int fdx_denorm_spec(fdx_handle h, const float* x, int B, int M, int T, const float* spec_min,
                    const float* spec_max, int n_spec, float* mel, fdx_stream s);
/* Standard-normal fill with the library's Philox4x32-10 generator (perf mode's stand-in for
 * torch.randn at diffusion.py:222). */
int fdx_randn(fdx_handle h, float* out, size_t n, uint64_t seed, uint64_t offset, fdx_stream s);

/* ------------------------------------------------------------------------------------------------
 * NSF-HiFiGAN generator -- replaces modules/vocoders/nsf_hifigan/models.py:353-448 (Generator),
 * :27-158 (ResBlock1/2), :161-350 (SineGen, SourceModuleHnNSF) and the scalar glue of
 * NsfHifiGAN.spec2wav, nsf_hifigan.py:72-85.
 * ---------------------------------------------------------------------------------------------- */
#define FDX_MAX_STAGES 8
#define FDX_MAX_RESK 4
#define FDX_MAX_DIL 4
typedef struct {
  int num_mels;                 /* 128 */
  int upsample_initial_channel; /* 512 */
  int n_stages;
  int upsample_rates[FDX_MAX_STAGES];
  int upsample_kernel_sizes[FDX_MAX_STAGES];
  int n_resblock_kernels;       /* 3 */
  int resblock_kernel_sizes[FDX_MAX_RESK];
  int n_dilations;              /* 3 for ResBlock1, 2 for ResBlock2 */
  int resblock_dilations[FDX_MAX_RESK][FDX_MAX_DIL];
  int resblock_type;            /* 1 | 2  (json "resblock") */
  int sampling_rate;            /* 44100 */
  int hop_size;                 /* prod(upsample_rates) */
  int harmonic_num;             /* 8 (models.py:360) */
} fdx_nsf_desc;

/* Canonical order (weight-norm already folded, nsf_hifigan.py:51-52): m_source.l_linear.{weight,bias},
 * conv_pre.{weight,bias}, then per stage i: ups.i.{weight,bias}, noise_convs.i.{weight,bias};
 * then per resblock n (stage-major): ResBlock1: per j: convs1.j.{weight,bias}, convs2.j.{weight,bias};
 * ResBlock2: per j: convs.j.{weight,bias}; finally conv_post.{weight,bias}.
 * Conv1d [Cout][Cin][k]; ConvTranspose1d [Cin][Cout][k]. */
int fdx_nsf_num_weights(const fdx_nsf_desc* d);
int fdx_nsf_packed_bytes(const fdx_nsf_desc* d, size_t* bytes);
int fdx_nsf_pack(const fdx_nsf_desc* d, const float* const* host_weights, int n_weights,
                 void* host_packed, size_t bytes);
int fdx_nsf_attach(fdx_handle h, const fdx_nsf_desc* d, const void* dev_packed, size_t bytes);
/* mel: dev [B][num_mels][T]; f0: dev [B][T] (Hz, 0 = unvoiced); mel_scale: 2.30259 when the mel is
 * log10 (nsf_hifigan.py:79-80) else 1.  rand_ini: dev [B][harmonic_num+1] in [0,1) (column 0 is forced
 * to 0 as models.py:213) or NULL; src_noise: dev [B][T*hop][harmonic_num+1] standard normals or NULL;
 * NULL => device Philox(seed).  wav: dev [B][T*hop]. */
int fdx_nsf_forward(fdx_handle h, const float* mel, const float* f0, int B, int T, float mel_scale,
                    const float* rand_ini, const float* src_noise, uint64_t seed, float* wav,
                    fdx_stream s);
/* Test hook: harmonic source only (models.py:411-416) -> har: dev [B][T*hop]. */
int fdx_nsf_source(fdx_handle h, const float* f0, int B, int T, const float* rand_ini,
                   const float* src_noise, uint64_t seed, float* har, fdx_stream s);

/* ------------------------------------------------------------------------------------------------
 * RefineGAN generator -- replaces modules/vocoders/refinegan/generator.py:313-478
 * (RefineGANGenerator), :14-83 (ResBlock), :86-107 (AdaIN), :110-156 (ParallelResBlock), :159-194
 * (CombToothGen) and the scalar glue of RefineGAN.spec2wav, refinegan/refinegan.py:67-78.
 * template_generator = "comb" only (the default; what hifi_svc_v2 / vocoder_refinegan configure).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int sampling_rate;            /* 44100 */
  int hop_length;               /* 256 = prod(downsample_rates) = prod(upsample_rates) */
  int n_down; int downsample_rates[FDX_MAX_STAGES];   /* (2, 2, 8, 8) */
  int n_up;   int upsample_rates[FDX_MAX_STAGES];     /* (8, 8, 2, 2): must mirror the downsample rates */
  int num_mels;                 /* 128 (256 inside HiFiSinger) */
  int start_channels;           /* 16 */
  float leaky_relu_slope;       /* 0.2 */
  int template_sine;            /* 0: template_generator="comb" (CombToothGen, generator.py:159-194); 1: "sine" (SineGen, :197-310) */
} fdx_refinegan_desc;
/* Canonical order = the module's state_dict order with weight norm folded: [template_sine: template_gen.merge.0.{weight,bias};] template_conv.{weight,bias};
 * per down stage i, per j<3: downsample_blocks.i.1.convs1.j.{weight,bias}, convs2.j.{weight,bias}; mel_conv.*;
 * source_conv.*; per up stage i: upsample_conv_blocks.i.input_conv.*, per branch b<3: blocks.b.0.weight,
 * per j<3: blocks.b.1.convs1.j.*, convs2.j.*, then blocks.b.2.weight; output_conv.*. */
int fdx_refinegan_num_weights(const fdx_refinegan_desc* d);
int fdx_refinegan_num_noises(const fdx_refinegan_desc* d);   /* 1 + 6 * n_up */
int fdx_refinegan_packed_bytes(const fdx_refinegan_desc* d, size_t* bytes);
int fdx_refinegan_pack(const fdx_refinegan_desc* d, const float* const* host_weights, int n_weights,
                       void* host_packed, size_t bytes);
int fdx_refinegan_attach(fdx_handle h, const fdx_refinegan_desc* d, const void* dev_packed, size_t bytes);
/* mel: dev [B][num_mels][T]; f0: dev [B][T]; wav: dev [B][T*hop_length].  noises: HOST array of
 * fdx_refinegan_num_noises() device pointers holding the standard-normal draws in the reference's order
 * ([B][1][L] comb noise, then per up stage / branch / (pre, post) AdaIN: [B][C][L_stage]), or NULL =>
 * device Philox(seed). */
int fdx_refinegan_forward(fdx_handle h, const float* mel, const float* f0, int B, int T, float mel_scale,
                          const float* const* noises, uint64_t seed, float* wav, fdx_stream s);

/* ------------------------------------------------------------------------------------------------
 * STFT / mel -- replaces utils/pitch_adjustable_mel.py:33-96 (PitchAdjustableMelSpectrogram.__call__),
 * utils/audio.py:11-18 (dynamic_range_compression) and the tail of NsfHifiGAN.wav2spec,
 * nsf_hifigan.py:101-107.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int sample_rate; /* 44100 */
  int n_fft;       /* 2048 */
  int win_size;    /* 2048 */
  int hop;         /* 512 */
  int n_mels;      /* 128 */
  float f_min;     /* 40 */
  float f_max;     /* 16000 */
} fdx_mel_desc;
int fdx_mel_config(fdx_handle h, const fdx_mel_desc* d);
/* number of frames the reference produces for N samples (center=False after the reflect pad). */
int fdx_mel_num_frames(const fdx_mel_desc* d, int N, float key_shift, float speed, int* T);
/* Pure host: the slaney mel filterbank [n_mels][1+n_fft/2] (librosa.filters.mel semantics). */
int fdx_mel_filterbank(const fdx_mel_desc* d, float* host_out);
enum { FDX_MEL_LINEAR = 0, FDX_MEL_LN = 1, FDX_MEL_LOG10 = 2 };
/* wav: dev [B][N]; mel: dev [B][n_mels][T] with T = fdx_mel_num_frames(...). */
int fdx_mel_forward(fdx_handle h, const float* wav, int B, int N, float key_shift, float speed,
                    int log_mode, float* mel, fdx_stream s);
/* The DFT matrix + window of every STFT geometry (key shift: pitch_adjustable_mel.py:34-37) are cached per handle and uploaded
 * asynchronously on the caller's stream: fdx_mel_forward never synchronises the stream unless more than 32 geometries are
 * alive (LRU eviction).  table_builds = geometries built so far, stream_syncs = synchronisations performed (0 in normal use),
 * cached = geometries held. */
int fdx_mel_stats(fdx_handle h, long* table_builds, long* stream_syncs, int* cached);

/* ------------------------------------------------------------------------------------------------
 * Condition front end -- replaces DiffSinger.forward_features,
 * archs/diffsinger/diffsinger.py:57-134, for the NaiveProjection encoders of the SVC configs
 * (modules/encoders/naive_projection.py:6-60; preprocessing utils/pitch.py:12-22).
 *   features[b][t][:] = W_text . contents[b][t][:] + b_text  (+ term_0) (+ term_1) ...   in the order given.
 * All pointers are device memory in the reference's own layouts (nn.Linear weight [out][in], nn.Embedding
 * weight [n][E]); nothing is packed.
 * ---------------------------------------------------------------------------------------------- */
enum { FDX_TERM_VECTOR = 0,        /* values: float [B][E] (per_frame = 0) or [B][T][E] (per_frame = 1): a float speaker mix */
       FDX_TERM_EMBEDDING = 1,     /* values: int64 ids [B]; w: table [n][E]                                              */
       FDX_TERM_SCALAR_LINEAR = 2  /* values: float [B] or [B][T]; adds w[E] * pre(value) + b[E]  (nn.Linear(1, E))       */ };
enum { FDX_PRE_NONE = 0, FDX_PRE_PITCH_TO_SCALE = 1 /* clamp((f0 - p0) / (p1 - p0), 0, 1), p0 = f0_min, p1 = f0_max */ };
#define FDX_MAX_FEATURE_TERMS 6
typedef struct {
  int kind, per_frame, preproc;
  int src_frames;   /* per_frame terms: frames the values tensor holds; 0 = T.  Otherwise frame t reads source frame
                       min(floor(t * (float)src_frames / T), src_frames - 1) -- repeat_expand, utils/tensor.py:7-43 */
  const void* values;
  const float* w;
  const float* b;   /* may be NULL */
  float p0, p1;
  /* NaiveProjectionEncoder(use_neck=True) (naive_projection.py:37-41) on a scalar term: Linear(1, neck) then Linear(neck, E):
     h[n] = neck_w[n] * pre(value) + neck_b[n], term = w[E][neck] . h + b[E].  neck = 0: the plain Linear(1, E) above. */
  int neck;
  const float* neck_w;
  const float* neck_b;   /* may be NULL */
} fdx_feature_term;
/* contents: dev [B][T][Din]; w_text: dev [E][Din]; b_text: dev [E] or NULL; terms: HOST array; features: dev [B][T][E]. */
int fdx_features_forward(fdx_handle h, const float* contents, int B, int T, int Din, int E, const float* w_text,
                         const float* b_text, const fdx_feature_term* terms, int n_terms, float* features, fdx_stream s);
/* Same launch with a trailing activation (nn.SiLU), an optional padding mask (dev [B][T] bytes, 1 => output 0) and an
 * optional channel-first output [B][E][T]: the two Linear + SiLU layers of HiFiSinger's feature_fuser and its
 * `features *= 1 - src_masks` (archs/hifisinger/core.py:24-29,109-110), writing what the generator consumes (:136-139). */
enum { FDX_ACT_NONE = 0, FDX_ACT_SILU = 1 };
int fdx_features_forward_ex(fdx_handle h, const float* contents, int B, int T, int Din, int E, const float* w_text,
                            const float* b_text, const fdx_feature_term* terms, int n_terms, int act,
                            const uint8_t* mask, int channel_first, float* features, fdx_stream s);

/* The same launch reading `contents` at the feature extractor's own frame rate and layout: contents is dev [B][S][Din], or
 * [B][Din][S] when contents_channel_first (what the extractors return, tools/diffusion/inference.py:113-114), and output
 * frame t uses source frame min(floor(t * (float)S / T), S - 1): `repeat_expand(text_features, mel_len).T`
 * (F.interpolate(mode="nearest"), utils/tensor.py:7-43) fused into the projection instead of materialising [T][Din]. */
int fdx_features_forward_src(fdx_handle h, const float* contents, int B, int S, int contents_channel_first, int T, int Din,
                             int E, const float* w_text, const float* b_text, const fdx_feature_term* terms, int n_terms,
                             int act, const uint8_t* mask, int channel_first, float* features, fdx_stream s);
/* The SVS variant of the same launch (archs/diffsinger/diffsinger.py:83-90; archs/hifisinger/core.py:71-79):
 *   phones2mel  dev int64 [B][T] or NULL: output frame t reads source frame phones2mel[b][t] (in [0, S)) instead of the
 *               nearest-expansion index -- torch.gather(text_encoder(contents), 1, phones2mel) -- and
 *   gather_mask dev bytes [B][T] or NULL: 1 => the gathered text features of that frame are multiplied by 0
 *               (`* (1 - mel_masks[:, :, None].float())`) BEFORE the additive terms;
 *   neck > 0    the text encoder is NaiveProjectionEncoder(use_neck=True) (naive_projection.py:37-41):
 *               Linear(Din, neck) [neck_w dev [neck][Din], neck_b dev [neck] or NULL] then Linear(neck, E) [w_text dev [E][neck]].
 *               neck <= FDX_MAX_NECK. */
#define FDX_MAX_NECK 32
int fdx_features_forward_svs(fdx_handle h, const float* contents, int B, int S, int contents_channel_first, int T, int Din,
                             int E, const float* w_text, const float* b_text, int neck, const float* neck_w, const float* neck_b,
                             const long long* phones2mel, const uint8_t* gather_mask, const fdx_feature_term* terms, int n_terms,
                             int act, const uint8_t* mask, int channel_first, float* features, fdx_stream s);
/* dst[r][t] = src[r][min(floor(t * (float)S / T), S - 1)] for r < rows: fish_diffusion.utils.tensor.repeat_expand
 * (mode "nearest") on its own, e.g. for a pitch track given at another frame rate (inference.py:108-109). */
int fdx_repeat_expand(fdx_handle h, const float* src, long rows, int S, int T, float* dst, fdx_stream s);

/* ------------------------------------------------------------------------------------------------
 * Kernel-level test / profiling hooks (used by tests/ and bench.py only)
 * ---------------------------------------------------------------------------------------------- */
/* y = conv1d(act_in(x), w) + bias with "same" zero padding: x dev [B][Cin][T], w HOST [Cout][Cin][k],
 * bias HOST [Cout] or NULL, y dev [B][Cout][T].  in_slope: leaky-relu slope applied to x (1 = none).
 * mode: 0 = 4-wave split-K tiles (denoiser regime), 1 = one tile per wave (vocoder regime). */
int fdx_debug_conv1d(fdx_handle h, const float* x, int B, int Cin, int T, const float* host_w,
                     const float* host_bias, int Cout, int k, int dilation, float in_slope, int mode,
                     float* y, fdx_stream s);
/* Per-kernel timing: when enabled, every launch of the dominant kernel (the dilated-conv + gate
 * kernel of the residual block) is dispatched with hipExtLaunchKernel's start/stop events, i.e. the
 * events carry that dispatch's own begin/end timestamps on its stream (the quantity rocprofv3's
 * kernel trace reports).  fdx_prof_read returns the number of launches recorded and their total
 * duration (synchronises those events). */
/* on = 0: off (and forget); 1: every launch; N > 1: every N-th launch (sampling keeps the probe effect negligible);
 * -1: pause -- stop recording but keep the recorded launches for fdx_prof_read. */
int fdx_prof_enable(fdx_handle h, int on);
/* Which kernel family fdx_prof_enable times: 0 = WaveNet dilated conv + gate (default), 1 = WaveNet out-projection + residual/skip,
 * 2 = NSF-HiFiGAN ResBlock convs (the no-split 64-row instantiation, i.e. the stages with >= 64 channels), 3 = the same instantiation
 * inside RefineGAN's ResBlocks, 4 = ConvNext pwconv1 (LayerNorm folded in, GELU), 5 = TransformerDecoder attention (self + cross).
 * fdx_prof_read's flops_per_launch is the mean algorithmic FLOP count of the recorded launches (they differ per stage for 2 / 3). */
int fdx_prof_select(fdx_handle h, int kind);
int fdx_prof_read(fdx_handle h, int* n_launches, double* total_ms, double* flops_per_launch);
/* The kernel instantiation (name as rocprofv3 prints it, MFMA instruction, workgroup tile) the launches recorded since the last
 * fdx_prof_enable actually ran -- bench.py reports this string instead of a literal, so the label cannot go stale when the
 * library picks another tile shape or another kernel family for the geometry.  Empty if nothing was recorded. */
int fdx_prof_label(fdx_handle h, char* buf, size_t cap);
/* Median elapsed time (ms) of an EMPTY hipEventRecord start/stop pair on stream s (diagnostic: what
 * bracketing a launch with plain recorded events would add; fdx_prof_* does not use that method). */
/* Recorded-sampler-graph cache of this handle: graphs captured so far, graph launches so far, graphs currently cached
 * (LRU of 48, FDX_GRAPH_CACHE=<n> overrides).  A serving loop in steady state shows launches growing and captures flat. */
int fdx_graph_stats(fdx_handle h, long* captures, long* launches, int* cached);
int fdx_prof_calibrate(fdx_handle h, fdx_stream s, double* empty_pair_ms);

#ifdef __cplusplus
}
#endif
#endif /* FISHDX_H_ */
