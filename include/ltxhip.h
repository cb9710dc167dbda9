/* ltxhip.h — C ABI of the MI355X-native LTX-Video denoise + decode engine.
 *
 * Drop-in boundary: the two Rust traits candle-video's `LtxPipeline` consumes as
 * `Box<dyn ...>` (reference: src/models/ltx_video/t2v_pipeline.rs)
 *     VideoTransformer3D  (:63-83)   ->  ltx_dit_*
 *     VaeLtxVideo         (:91-103)  ->  ltx_vae_*
 * plus the device-side pieces of `LtxPipeline::call` (:627-1073) and of
 * `FlowMatchEulerDiscreteScheduler` (scheduler.rs) that touch tensors
 *     guidance mix + rescale + Euler step  (t2v_pipeline.rs:941-964, 227-243; scheduler.rs:576-581)
 *     denormalize + decode-noise mix       (t2v_pipeline.rs:573-594, 1049-1062)
 * and a whole-pipeline entry (`ltx_pipeline_call`) that reproduces `LtxPipeline::call`
 * with text embeddings supplied (what examples/ltx-video/main.rs:620-646 does).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero error class otherwise;
 *     `ltx_last_error()` gives the thread-local message (reference: candle_core::Result / bail!).
 *   - tensor pointers are DEVICE pointers unless a parameter says "host".
 *   - the caller owns all tensors it passes; the library owns weights + workspaces.
 *   - one handle = one device; a handle is not re-entrant (reference: `&mut self`, single stream).
 *   - `stream` is a hipStream_t (NULL = default stream); all work is enqueued on it.  What blocks: workspace
 *     (re)allocation on a first/larger call and - unless ltx_warmup() / ltx_plan_load() / ltx_set_autotune(0) was
 *     used - the one-time measurement of the candidate GEMM plans on the first call of each GEMM shape.
 *   - results do not depend on the plans chosen (every plan of a shape sums K in the same order), so two processes
 *     given the same inputs produce the same bits whether or not they share a plan file.
 *   - model dtype: LTX_BF16 is the production path (bf16 storage, f32 accumulate — the reference's
 *     GPU dtype, main.rs:226); LTX_F32 is the parity path (exact f32 MFMA, reference CPU dtype).
 */
#ifndef LTXHIP_H
#define LTXHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { LTX_F32 = 0, LTX_BF16 = 1 } ltx_dtype;
typedef struct ltx_dit ltx_dit;
typedef struct ltx_vae ltx_vae;
typedef void* ltx_stream;   /* hipStream_t */

/* A named weight tensor as found in a safetensors file (host or device memory). */
typedef struct {
    const char* name;       /* diffusers-layout name, e.g. "transformer_blocks.0.attn1.to_q.weight" */
    const void* data;
    ltx_dtype dtype;        /* dtype of `data` */
    int ndim;
    int64_t shape[5];
    int on_device;          /* 0: host pointer, 1: device pointer */
} ltx_weight;

/* Mirrors LtxVideoTransformer3DModelConfig (ltx_transformer.rs:23-58). */
typedef struct {
    int in_channels, out_channels;
    int patch_size, patch_size_t;
    int num_attention_heads, attention_head_dim;
    int cross_attention_dim;
    int num_layers;
    float norm_eps;
    int caption_channels;
} ltx_dit_config;

/* Mirrors the decoder-side fields of AutoencoderKLLtxVideoConfig (vae.rs:32-103).  List fields are in config.json order
 * (the decoder constructor reverses them, vae.rs:1506-1519). */
typedef struct {
    int latent_channels, out_channels;
    int n_blocks;                        /* len(decoder_block_out_channels), <= 4 */
    int decoder_block_out_channels[4];
    int decoder_layers_per_block[5];     /* n_blocks + 1 entries */
    int decoder_upsample_factor[4];
    int patch_size, patch_size_t;
    int timestep_conditioning;
    int decoder_causal;
    float scaling_factor;
    int spatial_compression_ratio, temporal_compression_ratio;
    /* vae.rs:51, 676-690, 741-753: resnets of a flagged block add noise[h, w] * per_channel_scaleN[c] after conv1 / conv2 - where the
     * checkpoint holds "<resnet>.per_channel_scaleN.weight" (the reference's lookup; absent = no injection, as there).  n_blocks + 1
     * entries in the config's (encoder-side) order like the other lists.  The reference draws each [H, W] plane from the device RNG
     * (not reproducible); the engine draws plane k of a handle's life as Pcg32::new(seed, k).randn(H * W) - ltx_vae_set_noise_seed. */
    int decoder_inject_noise[5];
    /* vae.rs:52-53, 1103-1129, 1164-1168: whether an up-block's depth-to-space output gets the tiled-repeat residual. */
    int decoder_upsample_residual[4];
    /* vae.rs:40-41, 1212-1236: 1 = (2,2,2) upsampler; 0 = the spatial-only (1,2,2) form (frames kept, conv to 4 x channels);
     * temporal_compression_ratio is the caller's to set accordingly, as in the reference (vae.rs:2358-2434 reads the config). */
    int decoder_spatiotemporal_scaling[4];
    /* vae.rs:46-47: eps of the resnets' norm3 LayerNorm, which exists only where in_channels != out_channels (vae.rs:655-676)
     * - never in the decoder, whose resnets keep the channel count (vae.rs:1547-1549, 1242-1258).  Carried for
     * ltx_vae_get_config round trips; the decoder's RMS norms use the reference's hard-coded 1e-8 (vae.rs:620, 1575). */
    float resnet_eps;
} ltx_vae_config;

/* Tiling parameters of AutoencoderKLLtxVideo (vae.rs:1849-1861); NULL = untiled decode. */
typedef struct {
    int use_tiling, use_framewise_decoding;
    int tile_sample_min_height, tile_sample_min_width, tile_sample_min_num_frames;
    int tile_sample_stride_height, tile_sample_stride_width, tile_sample_stride_num_frames;
} ltx_tiling;

const char* ltx_last_error(void);
void ltx_dit_config_default(ltx_dit_config* cfg);      /* Default impl, ltx_transformer.rs:40-58 */
void ltx_vae_config_default(ltx_vae_config* cfg);      /* Default impl, vae.rs:68-103 */
void ltx_tiling_default(ltx_tiling* t);                /* vae.rs:1849-1861 */

/* ---- VideoTransformer3D (LtxVideoTransformer3DModel::new, ltx_transformer.rs:957-1022) ---- */
int ltx_dit_create(const ltx_dit_config* cfg, const ltx_weight* weights, size_t n_weights,
                   ltx_dtype model_dtype, int device, ltx_dit** out);
void ltx_dit_destroy(ltx_dit* m);
/* VideoTransformer3D::set_skip_block_list (t2v_pipeline.rs:82; ltx_transformer.rs:1024-1026); host ints */
int ltx_dit_set_skip_blocks(ltx_dit* m, const int* blocks, int n);
/* VideoTransformer3D::config (t2v_pipeline.rs:55-64) */
int ltx_dit_get_config(const ltx_dit* m, ltx_dit_config* out);
/* VideoTransformer3D::forward (t2v_pipeline.rs:68-80; ltx_transformer.rs:1029-1172).
 *   hidden   [B,S,in_channels]  io_dtype         enc  [B,K,caption_channels] io_dtype
 *   timestep [B] HOST f32                         enc_mask [B,K] f32 (1 keep / 0 pad) or NULL
 *   rope_scale: HOST float[3] or NULL             video_coords [B,S,3] f32 or NULL
 *   skip_layer_mask: HOST f32 [num_layers,B] or NULL (1 = skip)
 *   out      [B,S,out_channels] io_dtype
 * Any B >= 1 (batches beyond 8 rows run as chunks of 8: rows never interact in the forward). */
int ltx_dit_forward(ltx_dit* m, const void* hidden, const void* enc, const float* timestep,
                    const float* enc_mask, int B, int S, int K, int num_frames, int height, int width,
                    const float* rope_scale, const float* video_coords, const float* skip_layer_mask,
                    ltx_dtype io_dtype, void* out, ltx_stream stream);

/* Extension (no reference counterpart; results are identical): between enable=1 and enable=0 the caller promises
 * that an (enc pointer, enc_mask pointer, B, K) tuple identifies unchanged contents, so the caption projection
 * and the per-layer cross-attention K/V (which do not depend on timestep or latents, ltx_transformer.rs:1056,
 * 667-672) are computed once per tuple instead of once per forward; likewise a (video_coords pointer - or NULL -, B, S, grid,
 * rope_scale, stream) tuple identifies unchanged coordinates: the RoPE tables of the previous forward are kept while it repeats.
 * Buffers whose contents change inside the scope must change address (or the scope be closed).  ltx_pipeline_call uses it around its loop. */
int ltx_dit_context_cache(ltx_dit* m, int enable);

/* ---- VaeLtxVideo (AutoencoderKLLtxVideo::new decoder side, vae.rs:1765-1869) ---- */
/* weight names are the `decoder.*`, `latents_mean`, `latents_std` keys (vae.rs:1521-1608, 1827-1838) */
int ltx_vae_create(const ltx_vae_config* cfg, const ltx_weight* weights, size_t n_weights,
                   ltx_dtype model_dtype, int device, ltx_vae** out);
void ltx_vae_destroy(ltx_vae* v);
int ltx_vae_get_config(const ltx_vae* v, ltx_vae_config* out);
/* Noise injection (decoder_inject_noise): seed of the handle's plane stream, restarting the plane counter at 0.  Default seed 0.
 * A tiled decode of an injecting decoder runs its leaves one per decoder call (each leaf its own planes, as in the reference); a batch
 * beyond 8 samples is decoded 8 at a time, each such call with planes of its own (the reference: one plane per injection for the batch). */
int ltx_vae_set_noise_seed(ltx_vae* v, uint64_t seed);
/* 1 if any resnet of the decoder injects noise (flag set AND the scale present in the checkpoint), else 0 */
int ltx_vae_injects_noise(const ltx_vae* v);
const float* ltx_vae_latents_mean(const ltx_vae* v);   /* device f32 [latent_channels] */
const float* ltx_vae_latents_std(const ltx_vae* v);
/* VaeLtxVideo::decode (t2v_pipeline.rs:102; vae.rs:2101-2136, 2459-2462).
 *   latents  [B,C,F,H,W] io_dtype;  timestep HOST f32 [B] or NULL;  tiling NULL = direct decode
 *   out      [B,3,8F-7,32H,32W] f32, approx [-1,1]  (or [0,255] when postprocess == 1:
 *            LtxVideoProcessor::postprocess_video, t2v_pipeline.rs:146-155);
 *            postprocess == 2: `out` is a u8 buffer [B,8F-7,32H,32W,3] - the RGB8 frames the reference's CLI converts the
 *            post-processed tensor to (main.rs:653-675), written by conv_out's epilogue: the bytes of ltx_video_to_rgb8 on the
 *            postprocess == 1 result, a quarter of the bytes stored and no conversion pass */
int ltx_vae_decode(ltx_vae* v, const void* latents, ltx_dtype io_dtype, const float* timestep,
                   int B, int F, int H, int W, const ltx_tiling* tiling, int postprocess,
                   float* out, ltx_stream stream);
/* Same, but from packed tokens [B, F*H*W, C] f32 (the layout the denoise loop carries) with
 * unpack + denormalize + decode-noise mix fused in (t2v_pipeline.rs:1002-1067).
 *   noise: [B,C,F,H,W] f32 or NULL;  noise_scale HOST f32 [B] (ignored when noise NULL) */
int ltx_vae_decode_tokens(ltx_vae* v, const float* tokens, const float* noise, const float* noise_scale,
                          const float* timestep, int B, int F, int H, int W, const ltx_tiling* tiling,
                          int postprocess, float* out, ltx_stream stream);
/* Only the latent preparation of the call above (unpack is a view; denormalize with latents_mean/std and
 * scaling_factor, then (1-s)*z + s*noise, t2v_pipeline.rs:1002-1053), for callers that tile the decode themselves
 * (tile-sharded multi-GPU decode): out_tokens [B, F*H*W, C] f32, same token layout as the input. */
int ltx_vae_prepare_latents(ltx_vae* v, const float* tokens, const float* noise, const float* noise_scale,
                            int B, int F, int H, int W, float* out_tokens, ltx_stream stream);

/* ---- pipeline pieces that touch tensors ---- */
/* noise_pred = guidance(text, uncond?, perturbed?) ; latents += dt * noise_pred   (all f32 math)
 *   preds: [B, n] pred_dtype (uncond / perturbed may be NULL); latents [B, n] f32 in place (may be NULL);
 *   noise_pred_out optional f32 [B,n]; stats_ws: device workspace >= 32*B bytes (needed iff rescale > 0) */
int ltx_guidance_step(const void* text, const void* uncond, const void* perturbed, ltx_dtype pred_dtype,
                      float* latents, float* noise_pred_out, int B, int64_t n,
                      float guidance_scale, float guidance_rescale, float stg_scale, float dt,
                      void* stats_ws, ltx_stream stream);
/* Same guidance mix, followed by the scheduler's stochastic-sampling update (SchedulerConfig::stochastic_sampling,
 * scheduler.rs:557-575; the 0.9.6-distilled preset, configs.rs:210) instead of the Euler step:
 *   x0 = x - sigma * noise_pred;   x = (1 - sigma_next) * x0 + sigma_next * step_noise
 * step_noise [B, n] f32 is supplied by the caller (the reference draws it from the device RNG, :567). */
int ltx_guidance_step_stochastic(const void* text, const void* uncond, const void* perturbed, ltx_dtype pred_dtype,
                                 float* latents, float* noise_pred_out, int B, int64_t n,
                                 float guidance_scale, float guidance_rescale, float stg_scale,
                                 float sigma, float sigma_next, const float* step_noise,
                                 void* stats_ws, ltx_stream stream);

/* ---- host-side scalar restatements (no device work) ---- */
/* FlowMatchEulerDiscreteScheduler::set_timesteps via the Scheduler trait (scheduler.rs:274-412, 646-660).
 * sigmas_in: n values (custom list or linspace(1, 1/n, n)); writes n+1 sigmas and n truncated timesteps. */
int ltx_sched_set_timesteps(const float* sigmas_in, int n, float mu, int use_mu, float shift,
                            float shift_terminal, int use_shift_terminal, float* sigmas_out, int64_t* timesteps_out);
/* The same with the remaining options of FlowMatchEulerDiscreteSchedulerConfig (scheduler.rs:30-33): sigma_kind 0 = none,
 * 1 = use_karras_sigmas (convert_to_karras :222-235, rho 7), 2 = use_exponential_sigmas (:237-245), 3 = use_beta_sigmas
 * (:247-272, alpha = beta = 0.6, the inverse CDF of the Beta distribution in f64), applied after the stretch (:363-370);
 * invert_sigmas (:389-399): sigma -> 1 - sigma, timesteps from the inverted list, terminal sigma 1.  No preset of configs.rs
 * enables either. */
int ltx_sched_set_timesteps_ex(const float* sigmas_in, int n, float mu, int use_mu, float shift,
                               float shift_terminal, int use_shift_terminal, int sigma_kind, int invert_sigmas,
                               float* sigmas_out, int64_t* timesteps_out);
float ltx_calculate_shift(int seq_len, int base_seq_len, int max_seq_len, float base_shift, float max_shift);
/* Pcg32::new(seed, inc).randn(n) (utils/deterministic_rng.rs:11-81) into HOST memory */
int ltx_pcg32_randn(uint64_t seed, uint64_t inc, size_t n, float* out_host);
/* the first n values of Pcg32::new(seed, inc).next_u32() (deterministic_rng.rs:23-35) into HOST memory: bit-exact */
int ltx_pcg32_u32(uint64_t seed, uint64_t inc, size_t n, uint32_t* out_host);
/* video_coords of LtxPipeline::call (t2v_pipeline.rs:798-847) into HOST memory [F*H*W, 3] */
int ltx_build_video_coords(int F, int H, int W, int frame_rate, int ts_ratio, int sp_ratio, float* out_host);

/* ---- LtxPipeline::call (t2v_pipeline.rs:627-1073) with embeddings supplied ---- */
/* Per-step host hook: called on the calling thread before step `step` (0-based) of `num_steps` is enqueued, with the timestep the
 * model will see (LtxPipeline::current_timestep, t2v_pipeline.rs:865).  Non-zero return = interrupt: this and every later step is
 * skipped.  The hook must not call into the library on the same handles. */
typedef int (*ltx_step_fn)(void* user, int step, int num_steps, int64_t timestep);
typedef struct {
    int height, width, num_frames, frame_rate;
    int num_inference_steps;
    const float* sigmas;            /* HOST custom sigma list (len num_inference_steps) or NULL */
    float guidance_scale, guidance_rescale, stg_scale;
    const int* skip_block_list;     /* HOST */
    int n_skip_blocks;
    float decode_timestep, decode_noise_scale;
    int output_latent;              /* OutputType::Latent: stop before decode */
    int postprocess;                /* 1: apply postprocess_video; 2: that + RGB8 frames: out_video is then u8 [B,frames,height,width,3] (ltx_vae_decode) */
    const ltx_tiling* tiling;       /* NULL = untiled */
    float shift_terminal; int use_shift_terminal;   /* scheduler config (configs.rs:101-121) */
    int stochastic_sampling;        /* scheduler config (configs.rs:16; main.rs:550): stochastic step instead of Euler */
    const float* step_noise;        /* DEVICE f32 [num_inference_steps, B, S*C]: the per-step randn_like(sample) draws;
                                       required iff stochastic_sampling */
    const volatile int* interrupt;  /* HOST flag or NULL = LtxPipeline::interrupt (t2v_pipeline.rs:266, 861-863): read before every
                                       step; while non-zero the step is skipped (`continue`), the decode of the latents reached
                                       so far still runs, as in the reference */
    ltx_step_fn on_step;            /* NULL or the per-step hook above */
    void* on_step_user;
    /* With either of the two set, the host stays at most ONE step ahead of the device (it waits for step i-2 before polling for
     * step i), so a raised flag costs at most two more steps; without them the whole loop is enqueued without a wait. */
} ltx_pipeline_params;
void ltx_pipeline_params_default(ltx_pipeline_params* p);
/*   latents [B,S,128] f32 packed, updated in place;  prompt_embeds [B,K,4096] f32;  prompt_mask [B,K] f32;
 *   neg_* may be NULL when guidance_scale <= 1;  decode_noise [B,128,F,H,W] f32 or NULL;
 *   out_video [B,3,frames,height,width] f32 (ignored when output_latent) */
int ltx_pipeline_call(ltx_dit* dit, ltx_vae* vae, const ltx_pipeline_params* p,
                      float* latents, const float* prompt_embeds, const float* prompt_mask,
                      const float* neg_embeds, const float* neg_mask, const float* decode_noise,
                      int B, int K, float* out_video, ltx_stream stream);
/* per-stage wall time of the last ltx_pipeline_call on this thread, measured with hipEvents:
 * ms[0] = all DiT forwards, ms[1] = guidance+Euler, ms[2] = VAE decode (+denorm), ms[3] = total */
int ltx_pipeline_last_timing(float ms[4]);
/* denoise steps executed / requested by the last ltx_pipeline_call on this thread (they differ after an interrupt) */
int ltx_pipeline_last_steps(int* executed, int* requested);

/* ---- device memory for hosts without HIP bindings of their own (rust/hip_backend.rs keeps candle tensors on the CPU
 * device and moves the few MB per step itself; candle has no ROCm backend).  Copies are enqueued on `stream` and are
 * asynchronous with respect to the host for pinned memory only: call ltx_stream_synchronize before reading a d2h target. */
int ltx_device_alloc(size_t bytes, int device, void** out);
int ltx_device_free(void* p);
int ltx_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes, ltx_stream stream);
int ltx_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes, ltx_stream stream);
int ltx_stream_synchronize(ltx_stream stream);

/* ---- start-up control (no reference counterpart; speed only, never results) ----
 * ltx_warmup: run one forward (B, F*H*W tokens, K text tokens) and one decode of that latent geometry on scratch buffers,
 *   so that plan measurement, workspace sizing and code loading happen here and not inside the caller's first call.
 *   Either handle may be NULL.  Blocks until done.  A guided ltx_pipeline_call runs its guidance branches as one forward of
 *   branches x B rows (option guidance_batch): warm that geometry up with B = branches x batch (and the decode's with the batch itself).
 * ltx_set_autotune(0): never measure inside a call - shapes without a cached/loaded plan use the static cost model.
 * ltx_plan_save / ltx_plan_load: the measured plans of this process as a text file ("M N K conv ntaps T H W plan" per
 *   line); a loaded file makes a later process start with the same plans and without measuring. */
int ltx_warmup(ltx_dit* dit, ltx_vae* vae, int B, int F, int H, int W, int K, ltx_stream stream);
int ltx_set_autotune(int enabled);
int ltx_plan_save(const char* path);
int ltx_plan_load(const char* path);

/* ---- run-time options (A/B and diagnostic aids; the engine needs none of them) ----
 * One set of options per process: read ONCE, at first use, from the environment variable LTX_OPTIONS = "key=value,key=value",
 * changed with ltx_set_option (value NULL: the option's default), read back with ltx_get_option (the text ltx_set_option takes),
 * restored with ltx_reset_options (defaults + LTX_OPTIONS again).  Options are process state read by every launch WITHOUT a lock:
 * change them only while no other thread is inside a library call (tests and A/B drivers do; a serving process sets them once).
 * No launch path reads the environment.  The only other environment variable of the library is LTX_RCCL_LIB (path of the
 * RCCL library ltxhip_team.h loads).
 *
 *  speed only - the same bits with any value (PLANS never change a result: every plan of a GEMM shape sums K in one order):
 *   gemm_tune=0            no plan measurement: the static cost model (as ltx_set_autotune(0))
 *   gemm_plan=NAME         force one plan wherever a call is eligible for it: a gemm_big tile ("256x128", "160x256w16", ...),
 *                          "asm16:256x256" | "asm16:160x256" | "asm16:320x256" | "asm16" (its own tile choice), "ring:96x96" ... | "ring",
 *                          "p8:256" | "p8:128", "halo:128" | "halo:256"
 *   gemm_off=a+b           plan families left out of the choice: asm16, ring, p8, halo, halo_out (conv_out on the per-tap tile)
 *   gemm_wide_epi=0        fragment-wise 8-byte epilogue stores instead of the LDS-transposed 16-byte ones
 *   gemm_trace=1           print every bf16 GEMM shape left to the 128 x 128 register-staged kernel
 *   attn_q64_big=N         self-attention (head_dim 64): N 256-query blocks per head, the rest in 128-query blocks
 *   attn_q64_stream=1      self-attention (head_dim 64) as persistent workgroups that stream host-built item lists (key-range parts of
 *                          the last round's blocks merged in the launch: those rows round differently, <= 3e-3 rel-L2 from the block
 *                          grid; measured 4 - 6 % slower, kept as a tested option)
 *   norm_lean=0            the general RMS-norm map kernel instead of the DiT-specialised one
 *   vae_tile_batch=N       at most N leaves per decoder call of the tiled decode (-1: one)
 *   prof_kernel_events=0   stream-level event brackets in the ltx_prof_* timing
 *   ff2_defer=0            at <= 512 tokens the DiT's ff2 reduces its K ranges inside the launch instead of leaving them to the
 *                          row norm that follows (same partition and order either way)
 *  another ALGORITHM - results differ in rounding (each is a tested A/B arm against the oracle):
 *   gemm_off=big           every GEMM on the 128 x 128 kernel (small outputs then keep one K range)
 *   gemm_splitk=0          small outputs keep one K range (another f32 summation order)
 *   q2_fold=0 | 2          cross-attention q-norm as its own pass | folded whatever the shape
 *   norm_presum=0 | 2      row-reducing RMS norms | sums of squares from the producing GEMM whatever the shape
 *   norm_fold=0 | 1        the DiT's RMS norm + modulation between two GEMMs as its own pass | folded into the epilogues of the layer
 *                          that writes the rows (a second output h (1 + scale)) and of the layer that reads them (row 1 / rms, per-timestep
 *                          vector).  Default 2: the (1 + scale) factor in a per-timestep COPY of the reading layer's weights instead of
 *                          the second output (1.6 GB per distinct timestep at 2B, cached like the modulation; rows at different
 *                          timesteps, or a schedule with more distinct timesteps than norm_fold_copies (10), use form 1; a handle that
 *                          met such a schedule, or could not allocate a copy, frees its copies and stays on form 1)
 *   xattn_compact=0        cross attention multiplies every text key (differs from the default only for non-prefix masks)
 *   dense_qkv=0            q | k | v as column slices of one [M, 3D] matrix (same bits; another memory layout)
 *   vae_fuse_norm=0        the resnet's second norm as its own pass (1, default: fused where the conv's grid is about one round of
 *                          the chip or more; 2: fused on smaller grids too)
 *   t5_attn_mfma=0         the scalar T5 attention kernel
 *   guidance_batch=0       ltx_pipeline_call runs the guidance branches of a step (negative prompt / prompt / prompt with the STG
 *                          blocks skipped) as separate forwards, the reference's call order (default: one forward of up to 8 rows.
 *                          The same bits where B * S and branches * B * S rows select the same kernels and K partition - every shape
 *                          above 512 rows per branch, C2 / C3 / C5 among them; at a few hundred tokens the row-count-dependent
 *                          choices (split-K factor, one-row-per-block norms, deferred ff2) differ and the two forms agree to rounding)
 *   attn_off=a+b           attention kernels left out: q64, q128, cross, pipe (the next more general kernel serves)
 * Measured-negative experiments and tuning knobs ("x_name=int") exist only in builds made with -DLTX_EXPERIMENTS
 * (`make -C candle-video_amd experiments`, for tools/); the shipped library ignores them.  ltx_has_experiments() tells. */
int ltx_set_option(const char* key, const char* value);
int ltx_get_option(const char* key, char* out, int out_bytes);
int ltx_reset_options(void);
int ltx_has_experiments(void);

/* ---- optional measurement hooks (bench.py roofline object) ----
 * kinds: 0 linear GEMM, 1 conv3d implicit GEMM, 2 self-attention, 3 cross-attention, 4 row norms.
 * When enabled, every launch of those kernels is bracketed by hipEvents on ITS stream; report()
 * synchronises and returns the accumulated kernel time, algorithmic work (flops; bytes for kind 4)
 * and launch count since the last enable(). */
int ltx_prof_enable(int on);
int ltx_prof_report(int kind, double* total_ms, double* total_work, long long* count);
/* the same totals for ONE kernel inside a class: kernel 0 gemm_kernel (128 x 128), 1 gemm_big_kernel, 2 gemm_p8_kernel,
 * 3 conv_halo_kernel, 4 gemm_asm_kernel (32x32x16; experiment builds), 5 gemm_asm16_kernel, 6 gemm_ring_kernel; classes 2..4 have one kernel each
 * (index 0). */
int ltx_prof_report_kernel(int kind, int kernel, double* total_ms, double* total_work, long long* count);

#ifdef __cplusplus
}
#endif
#endif
