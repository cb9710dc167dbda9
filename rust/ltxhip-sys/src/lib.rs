//! Raw `extern "C"` bindings to `libltxhip.so` — one declaration per entry point of `include/ltxhip.h` that the
//! candle-video shim (`rust/hip_backend.rs`) uses, plus the checkpoint entry points of `include/ltxhip_weights.h`.
//!
//! Struct layouts are mirrored by hand in three places (the C header, the ctypes classes of
//! `candle-video_amd/ltxhip/__init__.py`, and here).  `tests/cabi_layout.c` prints `sizeof` / `offsetof` of the C
//! definitions; `tests/test_cabi_layout_cpu.py` compares them with the ctypes mirrors AND with the `LAYOUT_*` constants
//! below, which the `const _: () = assert!(..)` items tie to the Rust definitions at compile time.
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_float, c_int, c_void};
use core::mem::{align_of, size_of};

pub const LTX_F32: c_int = 0;
pub const LTX_BF16: c_int = 1;

/// `ltx_weight` (include/ltxhip.h): a named tensor as found in a safetensors file.
#[repr(C)]
pub struct ltx_weight {
    pub name: *const c_char,
    pub data: *const c_void,
    pub dtype: c_int,
    pub ndim: c_int,
    pub shape: [i64; 5],
    pub on_device: c_int,
}

/// `ltx_dit_config`: LtxVideoTransformer3DModelConfig (ltx_transformer.rs:23-58).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ltx_dit_config {
    pub in_channels: c_int,
    pub out_channels: c_int,
    pub patch_size: c_int,
    pub patch_size_t: c_int,
    pub num_attention_heads: c_int,
    pub attention_head_dim: c_int,
    pub cross_attention_dim: c_int,
    pub num_layers: c_int,
    pub norm_eps: c_float,
    pub caption_channels: c_int,
}

/// `ltx_vae_config`: decoder-side fields of AutoencoderKLLtxVideoConfig (vae.rs:32-103).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ltx_vae_config {
    pub latent_channels: c_int,
    pub out_channels: c_int,
    pub n_blocks: c_int,
    pub decoder_block_out_channels: [c_int; 4],
    pub decoder_layers_per_block: [c_int; 5],
    pub decoder_upsample_factor: [c_int; 4],
    pub patch_size: c_int,
    pub patch_size_t: c_int,
    pub timestep_conditioning: c_int,
    pub decoder_causal: c_int,
    pub scaling_factor: c_float,
    pub spatial_compression_ratio: c_int,
    pub temporal_compression_ratio: c_int,
    /// vae.rs:51 - any non-zero entry is refused (LTX_ERR_UNSUPPORTED); n_blocks + 1 entries, config.json order
    pub decoder_inject_noise: [c_int; 5],
    /// vae.rs:52-53 - tiled-repeat residual of each up-block's upsampler
    pub decoder_upsample_residual: [c_int; 4],
    /// vae.rs:40-41 - 0 = spatial-only up-block, refused (LTX_ERR_UNSUPPORTED)
    pub decoder_spatiotemporal_scaling: [c_int; 4],
    /// vae.rs:46-47 - norm3 only; unused by the decoder
    pub resnet_eps: c_float,
}

/// `ltx_tiling`: tiling parameters of AutoencoderKLLtxVideo (vae.rs:1849-1861).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ltx_tiling {
    pub use_tiling: c_int,
    pub use_framewise_decoding: c_int,
    pub tile_sample_min_height: c_int,
    pub tile_sample_min_width: c_int,
    pub tile_sample_min_num_frames: c_int,
    pub tile_sample_stride_height: c_int,
    pub tile_sample_stride_width: c_int,
    pub tile_sample_stride_num_frames: c_int,
}

/// `ltx_t5_config`: T5EncoderConfig (text_encoder.rs:66-113 / quantized_t5_encoder.rs:20-47); default = t5_xxl().
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ltx_t5_config {
    pub vocab_size: c_int,
    pub d_model: c_int,
    pub d_kv: c_int,
    pub d_ff: c_int,
    pub num_layers: c_int,
    pub num_heads: c_int,
    pub relative_attention_num_buckets: c_int,
    pub relative_attention_max_distance: c_int,
    pub layer_norm_epsilon: c_float,
}

/// `ltx_pipeline_params`: arguments of LtxPipeline::call (t2v_pipeline.rs:627-1073).
#[repr(C)]
pub struct ltx_pipeline_params {
    pub height: c_int,
    pub width: c_int,
    pub num_frames: c_int,
    pub frame_rate: c_int,
    pub num_inference_steps: c_int,
    pub sigmas: *const c_float,
    pub guidance_scale: c_float,
    pub guidance_rescale: c_float,
    pub stg_scale: c_float,
    pub skip_block_list: *const c_int,
    pub n_skip_blocks: c_int,
    pub decode_timestep: c_float,
    pub decode_noise_scale: c_float,
    pub output_latent: c_int,
    pub postprocess: c_int,
    pub tiling: *const ltx_tiling,
    pub shift_terminal: c_float,
    pub use_shift_terminal: c_int,
    pub stochastic_sampling: c_int,
    pub step_noise: *const c_float,
    pub interrupt: *const c_int,
    pub on_step: Option<unsafe extern "C" fn(user: *mut c_void, step: c_int, num_steps: c_int, timestep: i64) -> c_int>,
    pub on_step_user: *mut c_void,
}

// ---- layout guard: (size, align) per struct on LP64, compared with tests/cabi_layout.c by tests/test_cabi_layout_cpu.py ----
pub const LAYOUT_LTX_WEIGHT: (usize, usize) = (72, 8);
pub const LAYOUT_LTX_DIT_CONFIG: (usize, usize) = (40, 4);
pub const LAYOUT_LTX_VAE_CONFIG: (usize, usize) = (148, 4);
pub const LAYOUT_LTX_TILING: (usize, usize) = (32, 4);
pub const LAYOUT_LTX_PIPELINE_PARAMS: (usize, usize) = (136, 8);
pub const LAYOUT_LTX_T5_CONFIG: (usize, usize) = (36, 4);
const _: () = assert!(size_of::<ltx_weight>() == LAYOUT_LTX_WEIGHT.0 && align_of::<ltx_weight>() == LAYOUT_LTX_WEIGHT.1);
const _: () = assert!(size_of::<ltx_dit_config>() == LAYOUT_LTX_DIT_CONFIG.0 && align_of::<ltx_dit_config>() == LAYOUT_LTX_DIT_CONFIG.1);
const _: () = assert!(size_of::<ltx_vae_config>() == LAYOUT_LTX_VAE_CONFIG.0 && align_of::<ltx_vae_config>() == LAYOUT_LTX_VAE_CONFIG.1);
const _: () = assert!(size_of::<ltx_tiling>() == LAYOUT_LTX_TILING.0 && align_of::<ltx_tiling>() == LAYOUT_LTX_TILING.1);
const _: () = assert!(size_of::<ltx_pipeline_params>() == LAYOUT_LTX_PIPELINE_PARAMS.0 && align_of::<ltx_pipeline_params>() == LAYOUT_LTX_PIPELINE_PARAMS.1);

const _: () = assert!(size_of::<ltx_t5_config>() == LAYOUT_LTX_T5_CONFIG.0 && align_of::<ltx_t5_config>() == LAYOUT_LTX_T5_CONFIG.1);

/// opaque handles
pub enum ltx_dit {}
pub enum ltx_vae {}
pub enum ltx_t5 {}
/// hipStream_t; null = default stream
pub type ltx_stream = *mut c_void;

extern "C" {
    pub fn ltx_last_error() -> *const c_char;
    pub fn ltx_dit_config_default(cfg: *mut ltx_dit_config);
    pub fn ltx_vae_config_default(cfg: *mut ltx_vae_config);
    pub fn ltx_tiling_default(t: *mut ltx_tiling);

    // VideoTransformer3D (t2v_pipeline.rs:63-83)
    pub fn ltx_dit_create(cfg: *const ltx_dit_config, weights: *const ltx_weight, n_weights: usize, model_dtype: c_int, device: c_int, out: *mut *mut ltx_dit) -> c_int;
    pub fn ltx_dit_create_from_files(cfg: *const ltx_dit_config, path: *const c_char, unified: c_int, model_dtype: c_int, device: c_int, out: *mut *mut ltx_dit) -> c_int;
    pub fn ltx_dit_destroy(m: *mut ltx_dit);
    pub fn ltx_dit_get_config(m: *const ltx_dit, out: *mut ltx_dit_config) -> c_int;
    pub fn ltx_dit_set_skip_blocks(m: *mut ltx_dit, blocks: *const c_int, n: c_int) -> c_int;
    pub fn ltx_dit_context_cache(m: *mut ltx_dit, enable: c_int) -> c_int;
    pub fn ltx_dit_forward(m: *mut ltx_dit, hidden: *const c_void, enc: *const c_void, timestep: *const c_float, enc_mask: *const c_float,
        b: c_int, s: c_int, k: c_int, num_frames: c_int, height: c_int, width: c_int, rope_scale: *const c_float,
        video_coords: *const c_float, skip_layer_mask: *const c_float, io_dtype: c_int, out: *mut c_void, stream: ltx_stream) -> c_int;

    // VaeLtxVideo (t2v_pipeline.rs:91-103)
    pub fn ltx_vae_create(cfg: *const ltx_vae_config, weights: *const ltx_weight, n_weights: usize, model_dtype: c_int, device: c_int, out: *mut *mut ltx_vae) -> c_int;
    pub fn ltx_vae_config_from_json(json_path: *const c_char, cfg: *mut ltx_vae_config) -> c_int;
    pub fn ltx_vae_create_from_files(cfg: *const ltx_vae_config, path: *const c_char, unified: c_int, model_dtype: c_int, device: c_int, out: *mut *mut ltx_vae) -> c_int;
    pub fn ltx_vae_destroy(v: *mut ltx_vae);
    pub fn ltx_vae_get_config(v: *const ltx_vae, out: *mut ltx_vae_config) -> c_int;
    pub fn ltx_vae_set_noise_seed(v: *mut ltx_vae, seed: u64) -> c_int;
    pub fn ltx_vae_injects_noise(v: *const ltx_vae) -> c_int;
    pub fn ltx_vae_latents_mean(v: *const ltx_vae) -> *const c_float;
    pub fn ltx_vae_latents_std(v: *const ltx_vae) -> *const c_float;
    pub fn ltx_vae_decode(v: *mut ltx_vae, latents: *const c_void, io_dtype: c_int, timestep: *const c_float, b: c_int, f: c_int, h: c_int, w: c_int,
        tiling: *const ltx_tiling, postprocess: c_int, out: *mut c_float, stream: ltx_stream) -> c_int;

    // whole LtxPipeline::call on the device (t2v_pipeline.rs:627-1073)
    pub fn ltx_pipeline_params_default(p: *mut ltx_pipeline_params);
    pub fn ltx_pipeline_last_steps(executed: *mut c_int, requested: *mut c_int) -> c_int;
    pub fn ltx_get_option(key: *const c_char, out: *mut c_char, out_bytes: c_int) -> c_int;
    pub fn ltx_pipeline_call(dit: *mut ltx_dit, vae: *mut ltx_vae, p: *const ltx_pipeline_params, latents: *mut c_float, prompt_embeds: *const c_float,
        prompt_mask: *const c_float, neg_embeds: *const c_float, neg_mask: *const c_float, decode_noise: *const c_float, b: c_int, k: c_int,
        out_video: *mut c_float, stream: ltx_stream) -> c_int;

    // device memory + start-up control
    pub fn ltx_device_alloc(bytes: usize, device: c_int, out: *mut *mut c_void) -> c_int;
    pub fn ltx_device_free(p: *mut c_void) -> c_int;
    pub fn ltx_memcpy_h2d(dst_device: *mut c_void, src_host: *const c_void, bytes: usize, stream: ltx_stream) -> c_int;
    pub fn ltx_memcpy_d2h(dst_host: *mut c_void, src_device: *const c_void, bytes: usize, stream: ltx_stream) -> c_int;
    pub fn ltx_stream_synchronize(stream: ltx_stream) -> c_int;
    pub fn ltx_warmup(dit: *mut ltx_dit, vae: *mut ltx_vae, b: c_int, f: c_int, h: c_int, w: c_int, k: c_int, stream: ltx_stream) -> c_int;
    pub fn ltx_set_autotune(enabled: c_int) -> c_int;
    pub fn ltx_plan_save(path: *const c_char) -> c_int;
    pub fn ltx_plan_load(path: *const c_char) -> c_int;

    // text encoder (include/ltxhip_t5.h): T5TextEncoderWrapper (safetensors, bf16) and QuantizedT5EncoderModel (GGUF, masked)
    pub fn ltx_t5_config_default(c: *mut ltx_t5_config);
    pub fn ltx_t5_create(cfg: *const ltx_t5_config, weights: *const ltx_weight, n_weights: usize, model_dtype: c_int, device: c_int, out: *mut *mut ltx_t5) -> c_int;
    pub fn ltx_t5_create_from_gguf(cfg: *const ltx_t5_config, gguf_path: *const c_char, model_dtype: c_int, device: c_int, out: *mut *mut ltx_t5) -> c_int;
    pub fn ltx_t5_destroy(m: *mut ltx_t5);
    pub fn ltx_t5_forward(m: *mut ltx_t5, input_ids: *const i32, b: c_int, s: c_int, out_dtype: c_int, out: *mut c_void, stream: ltx_stream) -> c_int;
    pub fn ltx_t5_forward_masked(m: *mut ltx_t5, input_ids: *const i32, attention_mask: *const c_float, b: c_int, s: c_int, out_dtype: c_int,
                                 out: *mut c_void, stream: ltx_stream) -> c_int;

    // host-side scalar restatements
    pub fn ltx_pcg32_randn(seed: u64, inc: u64, n: usize, out_host: *mut c_float) -> c_int;
    pub fn ltx_pcg32_u32(seed: u64, inc: u64, n: usize, out_host: *mut u32) -> c_int;
}
