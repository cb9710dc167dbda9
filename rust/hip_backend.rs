//! candle-video backend for AMD Instinct MI355X: `impl VideoTransformer3D` / `impl VaeLtxVideo` over `libltxhip.so`.
//!
//! Drop this file into candle-video as `src/models/ltx_video/hip_backend.rs`, add `pub mod hip_backend;` to
//! `src/models/ltx_video/mod.rs`, add `ltxhip-sys = { path = ".../rust/ltxhip-sys", optional = true }` and a `hip`
//! feature to Cargo.toml (next to `cuda` / `flash-attn`, Cargo.toml:38-54), and box the two types where
//! `examples/ltx-video/main.rs:548-560` boxes `LtxVideoTransformer3DModel` / `AutoencoderKLLtxVideo`.
//! `LtxPipeline` itself (t2v_pipeline.rs:245-302, 627-1073) is untouched: it keeps driving the scheduler, the guidance
//! mix and the latent bookkeeping on candle tensors.
//!
//! candle has no ROCm device, so the pipeline's tensors live on `Device::Cpu`; the two heavy components keep their
//! weights and workspaces on the GPU and move only the call's operands: per denoise step 2.5 MB of latents in and out
//! plus (once, thanks to the context cache) the 2 MB of text embeddings; per video 458 MB of f32 frames out
//! (7.2 ms at PCIe Gen5 x16).  A caller that wants nothing on the host between steps uses `HipPipeline::call`, the
//! one-call form of `LtxPipeline::call` that keeps latents, predictions and the video in HBM (`ltx_pipeline_call`).
//!
//! STATUS: an UNTESTED SKETCH on the Rust side.  This file has never been compiled or type-checked (the authoring
//! environment has no Rust toolchain and no candle sources): the candle signatures it uses (`Tensor::from_vec`, `dims3`,
//! the trait imports) are written from the reference's own call sites and need a `cargo check` in a CI job that has a
//! toolchain before they are relied on.  What IS compiled and tested is everything below the `sys::` calls: the C ABI is
//! built, exported and exercised entry point by entry point through ctypes (tests/test_host_cabi.py, tests/test_gpu_*.py),
//! and the struct layouts and declared symbols of `ltxhip-sys` are checked against the headers and the library
//! (tests/test_cabi_layout_cpu.py).

use std::ffi::{c_void, CStr, CString};
use std::os::raw::c_int;
use std::path::Path;
use std::ptr;

use candle_core::{bail, DType, Device, Result, Tensor};
use ltxhip_sys as sys;

use super::t2v_pipeline::{TextEncoder as VTextEncoder, TransformerConfig, VaeConfig, VaeLtxVideo, VideoTransformer3D};

fn check(rc: c_int) -> Result<()> {
    if rc == 0 {
        return Ok(());
    }
    // same channel as the reference's `bail!` sites: a candle_core::Error carrying the library's message
    let msg = unsafe { CStr::from_ptr(sys::ltx_last_error()) }.to_string_lossy().into_owned();
    bail!("ltxhip (rc {rc}): {msg}")
}

fn model_dtype(dtype: DType) -> Result<c_int> {
    match dtype {
        DType::F32 => Ok(sys::LTX_F32),
        DType::BF16 => Ok(sys::LTX_BF16),
        other => bail!("ltxhip: model dtype must be F32 or BF16, got {other:?}"),
    }
}

/// A device allocation owned by the host side (grown on demand, freed on drop).
struct DeviceBuf {
    ptr: *mut c_void,
    bytes: usize,
    device: c_int,
}

impl DeviceBuf {
    fn new(device: c_int) -> Self {
        Self { ptr: ptr::null_mut(), bytes: 0, device }
    }
    fn ensure(&mut self, bytes: usize) -> Result<*mut c_void> {
        if bytes > self.bytes {
            if !self.ptr.is_null() {
                check(unsafe { sys::ltx_device_free(self.ptr) })?;
                self.ptr = ptr::null_mut();
                self.bytes = 0;
            }
            check(unsafe { sys::ltx_device_alloc(bytes, self.device, &mut self.ptr) })?;
            self.bytes = bytes;
        }
        Ok(self.ptr)
    }
    /// host f32 slice -> device
    fn upload(&mut self, data: &[f32]) -> Result<*const c_void> {
        let p = self.ensure(std::mem::size_of_val(data))?;
        check(unsafe { sys::ltx_memcpy_h2d(p, data.as_ptr() as *const c_void, std::mem::size_of_val(data), ptr::null_mut()) })?;
        Ok(p as *const c_void)
    }
}

impl Drop for DeviceBuf {
    fn drop(&mut self) {
        if !self.ptr.is_null() {
            unsafe { sys::ltx_device_free(self.ptr) };
        }
    }
}

/// Flattened f32 copy of a (CPU) candle tensor.
fn host_f32(t: &Tensor) -> Result<Vec<f32>> {
    t.to_device(&Device::Cpu)?.to_dtype(DType::F32)?.flatten_all()?.to_vec1::<f32>()
}

// ------------------------------------------------------------------------------------------------------------------
// VideoTransformer3D
// ------------------------------------------------------------------------------------------------------------------

/// Replaces `LtxVideoTransformer3DModel` (ltx_transformer.rs:955-1215) behind `VideoTransformer3D`.
pub struct HipDit {
    h: *mut sys::ltx_dit,
    cfg: TransformerConfig,
    out_channels: usize,
    hidden: DeviceBuf,
    coords: DeviceBuf,
    out: DeviceBuf,
    /// Text contexts currently on the device.  `LtxPipeline::call` alternates between the negative and the positive
    /// prompt under sequential CFG (t2v_pipeline.rs:878-939), so one slot would re-upload the context and drop the
    /// library's cached caption projection + cross-attention K/V on every forward; a few slots keep them all.  The
    /// library's cache keys on the device pointers, which stay the same for as long as a slot is not overwritten.
    ctx: Vec<CtxSlot>,
    ctx_clock: u64,
}

/// One uploaded (encoder_hidden_states, mask) pair.  The host copy is the identity of the contents (the pointer of a
/// candle storage would not survive a `to_dtype`).
struct CtxSlot {
    enc: DeviceBuf,
    mask: DeviceBuf,
    enc_host: Vec<f32>,
    mask_host: Vec<f32>,
    last_use: u64,
}

const CTX_SLOTS: usize = 4; // uncond + text + perturbed (same text) + one spare; the library keeps up to 4 contexts

impl HipDit {
    /// From a checkpoint on disk: `path` = the transformer's safetensors file or directory (`unified == false`,
    /// diffusers layout, loader.rs:319-330) or the Official unified file (`unified == true`, keys remapped as
    /// weight_format.rs:55-164) — what examples/ltx-video/main.rs:455-546 does with VarBuilder.
    pub fn from_files(cfg: &sys::ltx_dit_config, path: &Path, unified: bool, dtype: DType, device: usize) -> Result<Self> {
        let cpath = CString::new(path.to_string_lossy().as_bytes()).map_err(candle_core::Error::wrap)?;
        let mut h: *mut sys::ltx_dit = ptr::null_mut();
        check(unsafe { sys::ltx_dit_create_from_files(cfg, cpath.as_ptr(), unified as c_int, model_dtype(dtype)?, device as c_int, &mut h) })?;
        Ok(Self::wrap(h, cfg, device))
    }

    /// From tensors already in host memory (name -> candle tensor, diffusers names: ltx_transformer.rs:957-1003).
    pub fn new(cfg: &sys::ltx_dit_config, weights: &[(String, Tensor)], dtype: DType, device: usize) -> Result<Self> {
        let (names, datas, descs) = describe_weights(weights)?;
        let mut h: *mut sys::ltx_dit = ptr::null_mut();
        let rc = unsafe { sys::ltx_dit_create(cfg, descs.as_ptr(), descs.len(), model_dtype(dtype)?, device as c_int, &mut h) };
        drop((names, datas)); // kept alive until the library has copied them
        check(rc)?;
        Ok(Self::wrap(h, cfg, device))
    }

    fn wrap(h: *mut sys::ltx_dit, cfg: &sys::ltx_dit_config, device: usize) -> Self {
        let d = device as c_int;
        Self {
            h,
            cfg: TransformerConfig {
                in_channels: cfg.in_channels as usize,
                patch_size: cfg.patch_size as usize,
                patch_size_t: cfg.patch_size_t as usize,
                num_layers: cfg.num_layers as usize,
            },
            out_channels: cfg.out_channels as usize,
            hidden: DeviceBuf::new(d),
            coords: DeviceBuf::new(d),
            out: DeviceBuf::new(d),
            ctx: Vec::new(),
            ctx_clock: 0,
        }
    }

    /// Device pointers of the slot holding this context, uploading it (into a free or the least recently used slot) if new.
    fn context(&mut self, enc: Vec<f32>, mask: Vec<f32>) -> Result<(*const c_void, *const f32)> {
        self.ctx_clock += 1;
        let now = self.ctx_clock;
        if let Some(slot) = self.ctx.iter_mut().find(|c| c.enc_host == enc && c.mask_host == mask) {
            slot.last_use = now;
            return Ok((slot.enc.ptr as *const c_void, slot.mask.ptr as *const f32));
        }
        let d = self.hidden.device;
        if self.ctx.is_empty() {
            check(unsafe { sys::ltx_dit_context_cache(self.h, 1) })?; // first context: open the caching scope
        }
        let idx = if self.ctx.len() < CTX_SLOTS {
            self.ctx.push(CtxSlot { enc: DeviceBuf::new(d), mask: DeviceBuf::new(d), enc_host: Vec::new(), mask_host: Vec::new(), last_use: 0 });
            self.ctx.len() - 1
        } else {
            // a slot is overwritten in place: its device pointers now name other contents, so everything the library
            // cached is dropped (enable = 0 invalidates, enable = 1 re-opens the scope); the other slots refill on use
            check(unsafe { sys::ltx_dit_context_cache(self.h, 0) })?;
            check(unsafe { sys::ltx_dit_context_cache(self.h, 1) })?;
            (0..self.ctx.len()).min_by_key(|&i| self.ctx[i].last_use).unwrap()
        };
        let slot = &mut self.ctx[idx];
        slot.enc.upload(&enc)?;
        slot.mask.upload(&mask)?;
        slot.enc_host = enc;
        slot.mask_host = mask;
        slot.last_use = now;
        Ok((slot.enc.ptr as *const c_void, slot.mask.ptr as *const f32))
    }

    /// Default 2B configuration (ltx_transformer.rs:40-58).
    pub fn default_config() -> sys::ltx_dit_config {
        let mut c = std::mem::MaybeUninit::<sys::ltx_dit_config>::uninit();
        unsafe {
            sys::ltx_dit_config_default(c.as_mut_ptr());
            c.assume_init()
        }
    }

    pub(crate) fn handle(&self) -> *mut sys::ltx_dit {
        self.h
    }
}

impl Drop for HipDit {
    fn drop(&mut self) {
        unsafe { sys::ltx_dit_destroy(self.h) }
    }
}

impl VideoTransformer3D for HipDit {
    fn config(&self) -> &TransformerConfig {
        &self.cfg
    }

    fn set_skip_block_list(&mut self, list: Vec<usize>) {
        let v: Vec<c_int> = list.iter().map(|&x| x as c_int).collect();
        // the trait method cannot fail (t2v_pipeline.rs:82); a bad index is reported by the next forward
        unsafe { sys::ltx_dit_set_skip_blocks(self.h, v.as_ptr(), v.len() as c_int) };
    }

    #[allow(clippy::too_many_arguments)]
    fn forward(
        &mut self,
        hidden_states: &Tensor,
        encoder_hidden_states: &Tensor,
        timestep: &Tensor,
        encoder_attention_mask: &Tensor,
        num_frames: usize,
        height: usize,
        width: usize,
        rope_interpolation_scale: Option<(f32, f32, f32)>,
        video_coords: Option<&Tensor>,
        skip_layer_mask: Option<&Tensor>,
    ) -> Result<Tensor> {
        let (b, s, c_in) = hidden_states.dims3()?;
        let (_, k, _) = encoder_hidden_states.dims3()?;
        if c_in != self.cfg.in_channels {
            bail!("hidden_states has {c_in} channels, the model takes {}", self.cfg.in_channels);
        }
        let hidden = host_f32(hidden_states)?;
        let enc = host_f32(encoder_hidden_states)?;
        let mask = host_f32(encoder_attention_mask)?;
        let t = host_f32(timestep)?; // [B] host floats (ltx_transformer.rs:1051)
        if t.len() != b {
            bail!("timestep must have {b} entries, got {}", t.len());
        }
        let slm: Option<Vec<f32>> = skip_layer_mask.map(host_f32).transpose()?; // host [num_layers, B]
        let rs: Option<[f32; 3]> = rope_interpolation_scale.map(|r| [r.0, r.1, r.2]);

        let hidden_d = self.hidden.upload(&hidden)?;
        // the text contexts are step-invariant inside one LtxPipeline::call: each is uploaded once and the library keeps
        // its caption projection + cross-attention K/V for as long as the slot's device pointers keep their contents
        let (enc_d, mask_d) = self.context(enc, mask)?;
        let coords_d = match video_coords {
            Some(c) => self.coords.upload(&host_f32(c)?)? as *const f32,
            None => ptr::null(),
        };
        let out_elems = b * s * self.out_channels;
        let out_d = self.out.ensure(out_elems * 4)?;
        check(unsafe {
            sys::ltx_dit_forward(
                self.h,
                hidden_d,
                enc_d,
                t.as_ptr(),
                mask_d,
                b as c_int,
                s as c_int,
                k as c_int,
                num_frames as c_int,
                height as c_int,
                width as c_int,
                rs.as_ref().map_or(ptr::null(), |r| r.as_ptr()),
                coords_d,
                slm.as_ref().map_or(ptr::null(), |v| v.as_ptr()),
                sys::LTX_F32,
                out_d,
                ptr::null_mut(),
            )
        })?;
        let mut host = vec![0f32; out_elems];
        check(unsafe { sys::ltx_memcpy_d2h(host.as_mut_ptr() as *mut c_void, out_d, out_elems * 4, ptr::null_mut()) })?;
        check(unsafe { sys::ltx_stream_synchronize(ptr::null_mut()) })?;
        // the reference returns the model dtype and the pipeline casts to f32 (t2v_pipeline.rs:942, 984): f32 is returned here
        Tensor::from_vec(host, (b, s, self.out_channels), hidden_states.device())
    }
}

// ------------------------------------------------------------------------------------------------------------------
// VaeLtxVideo
// ------------------------------------------------------------------------------------------------------------------

/// Replaces `AutoencoderKLLtxVideo` (vae.rs:1729-2463) behind `VaeLtxVideo` (decode side).
pub struct HipVae {
    h: *mut sys::ltx_vae,
    cfg: VaeConfig,
    ccfg: sys::ltx_vae_config,
    mean: Tensor,
    std: Tensor,
    /// vae.rs:1744-1758 (off unless asked, main.rs:514-516)
    pub use_tiling: bool,
    pub use_framewise_decoding: bool,
    pub tiling: sys::ltx_tiling,
    device: c_int,
}

impl HipVae {
    pub fn from_files(cfg: &sys::ltx_vae_config, path: &Path, unified: bool, dtype: DType, device: usize) -> Result<Self> {
        let cpath = CString::new(path.to_string_lossy().as_bytes()).map_err(candle_core::Error::wrap)?;
        let mut h: *mut sys::ltx_vae = ptr::null_mut();
        check(unsafe { sys::ltx_vae_create_from_files(cfg, cpath.as_ptr(), unified as c_int, model_dtype(dtype)?, device as c_int, &mut h) })?;
        // a vae/config.json beside diffusers-layout weights replaces `cfg` (main.rs:525-534): ask the engine what it built
        let mut eff = *cfg;
        check(unsafe { sys::ltx_vae_get_config(h, &mut eff) })?;
        Self::wrap(h, &eff, device)
    }

    /// weights: the `decoder.*`, `latents_mean`, `latents_std` keys (vae.rs:1521-1608, 1827-1838)
    pub fn new(cfg: &sys::ltx_vae_config, weights: &[(String, Tensor)], dtype: DType, device: usize) -> Result<Self> {
        let (names, datas, descs) = describe_weights(weights)?;
        let mut h: *mut sys::ltx_vae = ptr::null_mut();
        let rc = unsafe { sys::ltx_vae_create(cfg, descs.as_ptr(), descs.len(), model_dtype(dtype)?, device as c_int, &mut h) };
        drop((names, datas));
        check(rc)?;
        Self::wrap(h, cfg, device)
    }

    fn wrap(h: *mut sys::ltx_vae, cfg: &sys::ltx_vae_config, device: usize) -> Result<Self> {
        let c = cfg.latent_channels as usize;
        let fetch = |p: *const f32| -> Result<Tensor> {
            let mut host = vec![0f32; c];
            check(unsafe { sys::ltx_memcpy_d2h(host.as_mut_ptr() as *mut c_void, p as *const c_void, c * 4, ptr::null_mut()) })?;
            check(unsafe { sys::ltx_stream_synchronize(ptr::null_mut()) })?;
            Tensor::from_vec(host, c, &Device::Cpu)
        };
        let mean = fetch(unsafe { sys::ltx_vae_latents_mean(h) })?;
        let std = fetch(unsafe { sys::ltx_vae_latents_std(h) })?;
        let mut tiling = std::mem::MaybeUninit::<sys::ltx_tiling>::uninit();
        let tiling = unsafe {
            sys::ltx_tiling_default(tiling.as_mut_ptr());
            tiling.assume_init()
        };
        Ok(Self {
            h,
            cfg: VaeConfig { scaling_factor: cfg.scaling_factor, timestep_conditioning: cfg.timestep_conditioning != 0 },
            ccfg: *cfg,
            mean,
            std,
            use_tiling: false,
            use_framewise_decoding: false,
            tiling,
            device: device as c_int,
        })
    }

    pub fn default_config() -> sys::ltx_vae_config {
        let mut c = std::mem::MaybeUninit::<sys::ltx_vae_config>::uninit();
        unsafe {
            sys::ltx_vae_config_default(c.as_mut_ptr());
            c.assume_init()
        }
    }

    pub(crate) fn handle(&self) -> *mut sys::ltx_vae {
        self.h
    }
}

impl Drop for HipVae {
    fn drop(&mut self) {
        unsafe { sys::ltx_vae_destroy(self.h) }
    }
}

impl VaeLtxVideo for HipVae {
    fn dtype(&self) -> DType {
        DType::F32 // decode() takes f32 latents and returns f32 frames, whatever the model dtype
    }
    fn spatial_compression_ratio(&self) -> usize {
        self.ccfg.spatial_compression_ratio as usize
    }
    fn temporal_compression_ratio(&self) -> usize {
        self.ccfg.temporal_compression_ratio as usize
    }
    fn config(&self) -> &VaeConfig {
        &self.cfg
    }
    fn latents_mean(&self) -> &Tensor {
        &self.mean
    }
    fn latents_std(&self) -> &Tensor {
        &self.std
    }

    fn decode(&self, latents: &Tensor, timestep: Option<&Tensor>) -> Result<Tensor> {
        let (b, c, f, h, w) = latents.dims5()?;
        if c != self.ccfg.latent_channels as usize {
            bail!("latents have {c} channels, the decoder takes {}", self.ccfg.latent_channels);
        }
        let z = host_f32(latents)?;
        let t: Option<Vec<f32>> = timestep.map(host_f32).transpose()?;
        let tr = self.temporal_compression_ratio();
        let sr = self.spatial_compression_ratio();
        let (fo, ho, wo) = ((f - 1) * tr + 1, h * sr, w * sr); // vae.rs:2101-2136
        let n_out = b * 3 * fo * ho * wo;
        let mut zin = DeviceBuf::new(self.device);
        let mut vout = DeviceBuf::new(self.device);
        let z_d = zin.upload(&z)?;
        let v_d = vout.ensure(n_out * 4)?;
        let mut tl = self.tiling;
        tl.use_tiling = self.use_tiling as c_int;
        tl.use_framewise_decoding = self.use_framewise_decoding as c_int;
        let tl_ptr = if self.use_tiling || self.use_framewise_decoding { &tl as *const sys::ltx_tiling } else { ptr::null() };
        check(unsafe {
            sys::ltx_vae_decode(
                self.h,
                z_d,
                sys::LTX_F32,
                t.as_ref().map_or(ptr::null(), |v| v.as_ptr()),
                b as c_int,
                f as c_int,
                h as c_int,
                w as c_int,
                tl_ptr,
                0, // raw [-1, 1] frames: LtxVideoProcessor::postprocess_video stays with the pipeline (t2v_pipeline.rs:146-155)
                v_d as *mut f32,
                ptr::null_mut(),
            )
        })?;
        let mut host = vec![0f32; n_out];
        check(unsafe { sys::ltx_memcpy_d2h(host.as_mut_ptr() as *mut c_void, v_d, n_out * 4, ptr::null_mut()) })?;
        check(unsafe { sys::ltx_stream_synchronize(ptr::null_mut()) })?;
        Tensor::from_vec(host, (b, 3, fo, ho, wo), latents.device())
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Text encoder
// ------------------------------------------------------------------------------------------------------------------

/// Replaces `QuantizedT5EncoderModel` (quantized_t5_encoder.rs:558-679, the DEFAULT of examples/ltx-video/main.rs:441-444)
/// and, through `VTextEncoder`, `T5TextEncoderWrapper` (text_encoder.rs:596-606).
pub struct HipT5 {
    handle: *mut sys::ltx_t5,
    dtype: DType,
    d_model: usize,
    out: DeviceBuf,
}

impl HipT5 {
    /// `QuantizedT5EncoderModel::load(gguf_path, device)`: T5-XXL from a GGUF file, every tensor dequantised on the GPU.
    /// `DType::F32` is the reference's arithmetic (QLinear dequantises to f32); `BF16` rounds the dequantised weights once.
    pub fn load(gguf_path: &Path, dtype: DType, device: c_int) -> Result<Self> {
        let mut cfg = std::mem::MaybeUninit::<sys::ltx_t5_config>::uninit();
        unsafe { sys::ltx_t5_config_default(cfg.as_mut_ptr()) };
        Self::load_with_config(gguf_path, unsafe { cfg.assume_init() }, dtype, device)
    }

    pub fn load_with_config(gguf_path: &Path, cfg: sys::ltx_t5_config, dtype: DType, device: c_int) -> Result<Self> {
        let path = CString::new(gguf_path.to_string_lossy().as_bytes()).map_err(candle_core::Error::wrap)?;
        let mut handle = ptr::null_mut();
        check(unsafe { sys::ltx_t5_create_from_gguf(&cfg, path.as_ptr(), model_dtype(dtype)?, device, &mut handle) })?;
        Ok(Self { handle, dtype, d_model: cfg.d_model as usize, out: DeviceBuf::new(device) })
    }

    /// `forward(input_ids [B,S] u32, Some(mask [B,S]))` (:608-650) -> [B,S,d_model] f32 on `Device::Cpu`.
    pub fn forward_masked(&mut self, input_ids: &Tensor, attention_mask: Option<&Tensor>) -> Result<Tensor> {
        let (b, s) = input_ids.dims2()?;
        let ids: Vec<i32> = input_ids.to_device(&Device::Cpu)?.to_dtype(DType::U32)?.flatten_all()?.to_vec1::<u32>()?.into_iter().map(|x| x as i32).collect();
        let mask = match attention_mask { Some(m) => Some(host_f32(m)?), None => None };
        let n = b * s * self.d_model;
        let dev = self.out.ensure(n * 4)?;
        check(unsafe {
            sys::ltx_t5_forward_masked(self.handle, ids.as_ptr(), mask.as_ref().map_or(ptr::null(), |m| m.as_ptr()), b as c_int, s as c_int,
                                       sys::LTX_F32, dev, ptr::null_mut())
        })?;
        let mut host = vec![0f32; n];
        check(unsafe { sys::ltx_memcpy_d2h(host.as_mut_ptr() as *mut c_void, dev, n * 4, ptr::null_mut()) })?;
        check(unsafe { sys::ltx_stream_synchronize(ptr::null_mut()) })?;
        Tensor::from_vec(host, (b, s, self.d_model), &Device::Cpu)
    }
}

impl Drop for HipT5 {
    fn drop(&mut self) {
        unsafe { sys::ltx_t5_destroy(self.handle) };
    }
}

impl VTextEncoder for HipT5 {
    fn dtype(&self) -> DType {
        self.dtype
    }
    fn forward(&mut self, input_ids: &Tensor) -> Result<Tensor> {
        self.forward_masked(input_ids, None)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// one-call form: LtxPipeline::call with everything resident in HBM
// ------------------------------------------------------------------------------------------------------------------

/// Arguments of `LtxPipeline::call` that reach the device entry (t2v_pipeline.rs:627-690).
pub struct HipCall<'a> {
    pub height: usize,
    pub width: usize,
    pub num_frames: usize,
    pub frame_rate: usize,
    pub num_inference_steps: usize,
    pub sigmas: Option<&'a [f32]>,
    pub guidance_scale: f32,
    pub guidance_rescale: f32,
    pub stg_scale: f32,
    pub skip_block_list: Option<&'a [usize]>,
    pub decode_timestep: f32,
    pub decode_noise_scale: f32,
}

pub struct HipPipeline<'a> {
    pub dit: &'a mut HipDit,
    pub vae: &'a HipVae,
}

impl<'a> HipPipeline<'a> {
    /// latents [B, F*H*W, 128] f32 packed; prompt_embeds [B, K, 4096]; masks [B, K]; decode_noise [B,128,F,H,W] or None.
    /// Returns the post-processed video [B, 3, frames, height, width] in [0, 255] (t2v_pipeline.rs:146-155, 1064-1070).
    pub fn call(&mut self, a: &HipCall, latents: &Tensor, prompt_embeds: &Tensor, prompt_mask: &Tensor,
                neg: Option<(&Tensor, &Tensor)>, decode_noise: Option<&Tensor>) -> Result<Tensor> {
        let (b, _s, _c) = latents.dims3()?;
        let (_, k, _) = prompt_embeds.dims3()?;
        let dev = self.vae.device;
        let mut bufs: Vec<DeviceBuf> = Vec::new();
        let mut up = |t: &Tensor| -> Result<*const f32> {
            let mut d = DeviceBuf::new(dev);
            let p = d.upload(&host_f32(t)?)? as *const f32;
            bufs.push(d);
            Ok(p)
        };
        let lat_d = up(latents)? as *mut f32;
        let pe_d = up(prompt_embeds)?;
        let pm_d = up(prompt_mask)?;
        let (ne_d, nm_d) = match neg {
            Some((e, m)) => (up(e)?, up(m)?),
            None => (ptr::null(), ptr::null()),
        };
        let dn_d = match decode_noise {
            Some(n) => up(n)?,
            None => ptr::null(),
        };
        let skip: Vec<c_int> = a.skip_block_list.map(|l| l.iter().map(|&x| x as c_int).collect()).unwrap_or_default();
        let mut p = std::mem::MaybeUninit::<sys::ltx_pipeline_params>::uninit();
        let mut p = unsafe {
            sys::ltx_pipeline_params_default(p.as_mut_ptr());
            p.assume_init()
        };
        p.height = a.height as c_int;
        p.width = a.width as c_int;
        p.num_frames = a.num_frames as c_int;
        p.frame_rate = a.frame_rate as c_int;
        p.num_inference_steps = a.num_inference_steps as c_int;
        p.sigmas = a.sigmas.map_or(ptr::null(), |s| s.as_ptr());
        p.guidance_scale = a.guidance_scale;
        p.guidance_rescale = a.guidance_rescale;
        p.stg_scale = a.stg_scale;
        if a.skip_block_list.is_some() {
            p.skip_block_list = skip.as_ptr();
            p.n_skip_blocks = skip.len() as c_int;
        }
        p.decode_timestep = a.decode_timestep;
        p.decode_noise_scale = a.decode_noise_scale;
        p.postprocess = 1;
        let n_out = b * 3 * a.num_frames * a.height * a.width;
        let mut vout = DeviceBuf::new(dev);
        let v_d = vout.ensure(n_out * 4)? as *mut f32;
        check(unsafe { sys::ltx_pipeline_call(self.dit.handle(), self.vae.handle(), &p, lat_d, pe_d, pm_d, ne_d, nm_d, dn_d, b as c_int, k as c_int, v_d, ptr::null_mut()) })?;
        let mut host = vec![0f32; n_out];
        check(unsafe { sys::ltx_memcpy_d2h(host.as_mut_ptr() as *mut c_void, v_d as *const c_void, n_out * 4, ptr::null_mut()) })?;
        check(unsafe { sys::ltx_stream_synchronize(ptr::null_mut()) })?;
        Tensor::from_vec(host, (b, 3, a.num_frames, a.height, a.width), &Device::Cpu)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// weights handed over from host tensors
// ------------------------------------------------------------------------------------------------------------------

type Described = (Vec<CString>, Vec<Vec<f32>>, Vec<sys::ltx_weight>);

/// `(name, tensor)` pairs -> `ltx_weight` descriptors over host f32 copies (kept alive by the returned vectors).
fn describe_weights(weights: &[(String, Tensor)]) -> Result<Described> {
    let mut names = Vec::with_capacity(weights.len());
    let mut datas = Vec::with_capacity(weights.len());
    let mut descs = Vec::with_capacity(weights.len());
    for (name, t) in weights {
        let dims = t.dims();
        if dims.len() > 5 {
            bail!("weight {name} has {} dimensions (max 5)", dims.len());
        }
        let mut shape = [1i64; 5];
        for (i, &d) in dims.iter().enumerate() {
            shape[i] = d as i64;
        }
        names.push(CString::new(name.as_bytes()).map_err(candle_core::Error::wrap)?);
        datas.push(host_f32(t)?);
        descs.push(sys::ltx_weight {
            name: names.last().unwrap().as_ptr(),
            data: datas.last().unwrap().as_ptr() as *const c_void,
            dtype: sys::LTX_F32,
            ndim: dims.len() as c_int,
            shape,
            on_device: 0,
        });
    }
    Ok((names, datas, descs))
}
