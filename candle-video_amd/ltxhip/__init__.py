"""Host-side mirror of candle-video's LTX pipeline interface over libltxhip.so (C ABI).

Classes and argument meaning follow the reference (FerrisMind/candle-video,
src/models/ltx_video/):

    LtxVideoTransformer3DModel.forward      ltx_transformer.rs:1029-1172 / trait t2v_pipeline.rs:63-83
    AutoencoderKLLtxVideo.decode            vae.rs:2101-2136 / trait t2v_pipeline.rs:91-103
    FlowMatchEulerDiscreteScheduler         scheduler.rs:646-668 (Scheduler trait)
    LtxPipeline.call                        t2v_pipeline.rs:627-1073

torch is used ONLY as the device-memory / stream provider (tensors in, tensors
out); every computation happens in the HIP library.  There is no CPU fallback:
importing this module without the built library raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LTXHIP_LIB: a variant build of the SAME library (tools/variants/libltxhip_*.so, the A/B tooling); never a fallback
_LIB_PATH = os.environ.get("LTXHIP_LIB") or os.path.join(os.path.dirname(_HERE), "libltxhip.so")

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"ltxhip: {_LIB_PATH} is missing — build it with `make -C candle-video_amd` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
lib = C.CDLL(_LIB_PATH)

LTX_F32, LTX_BF16 = 0, 1


class LtxError(RuntimeError):
    pass


lib.ltx_last_error.restype = C.c_char_p


def _check(rc: int):
    if rc != 0:
        raise LtxError(f"[ltxhip rc={rc}] {lib.ltx_last_error().decode()}")


class _Weight(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int), ("ndim", C.c_int),
                ("shape", C.c_int64 * 5), ("on_device", C.c_int)]


class DitConfigC(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("in_channels", "out_channels", "patch_size", "patch_size_t",
                                       "num_attention_heads", "attention_head_dim", "cross_attention_dim", "num_layers")] + \
               [("norm_eps", C.c_float), ("caption_channels", C.c_int)]


class VaeConfigC(C.Structure):
    _fields_ = [("latent_channels", C.c_int), ("out_channels", C.c_int), ("n_blocks", C.c_int),
                ("decoder_block_out_channels", C.c_int * 4), ("decoder_layers_per_block", C.c_int * 5),
                ("decoder_upsample_factor", C.c_int * 4), ("patch_size", C.c_int), ("patch_size_t", C.c_int),
                ("timestep_conditioning", C.c_int), ("decoder_causal", C.c_int), ("scaling_factor", C.c_float),
                ("spatial_compression_ratio", C.c_int), ("temporal_compression_ratio", C.c_int),
                ("decoder_inject_noise", C.c_int * 5), ("decoder_upsample_residual", C.c_int * 4),
                ("decoder_spatiotemporal_scaling", C.c_int * 4), ("resnet_eps", C.c_float)]


class TilingC(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("use_tiling", "use_framewise_decoding", "tile_sample_min_height", "tile_sample_min_width",
                                       "tile_sample_min_num_frames", "tile_sample_stride_height", "tile_sample_stride_width",
                                       "tile_sample_stride_num_frames")]


class PipelineParamsC(C.Structure):
    _fields_ = [("height", C.c_int), ("width", C.c_int), ("num_frames", C.c_int), ("frame_rate", C.c_int),
                ("num_inference_steps", C.c_int), ("sigmas", C.POINTER(C.c_float)),
                ("guidance_scale", C.c_float), ("guidance_rescale", C.c_float), ("stg_scale", C.c_float),
                ("skip_block_list", C.POINTER(C.c_int)), ("n_skip_blocks", C.c_int),
                ("decode_timestep", C.c_float), ("decode_noise_scale", C.c_float),
                ("output_latent", C.c_int), ("postprocess", C.c_int), ("tiling", C.POINTER(TilingC)),
                ("shift_terminal", C.c_float), ("use_shift_terminal", C.c_int),
                ("stochastic_sampling", C.c_int), ("step_noise", C.POINTER(C.c_float)),
                ("interrupt", C.POINTER(C.c_int)), ("on_step", C.c_void_p), ("on_step_user", C.c_void_p)]


StepFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64)     # ltx_step_fn


_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_fp = C.POINTER(C.c_float)
_SIGS = {
    "ltx_dit_config_default": [_vp], "ltx_vae_config_default": [_vp], "ltx_tiling_default": [_vp],
    "ltx_dit_create": [_vp, _vp, _sz, _i, _i, _vp], "ltx_dit_destroy": [_vp],
    "ltx_dit_set_skip_blocks": [_vp, _vp, _i], "ltx_dit_context_cache": [_vp, _i], "ltx_dit_get_config": [_vp, _vp],
    "ltx_dit_forward": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "ltx_vae_create": [_vp, _vp, _sz, _i, _i, _vp], "ltx_vae_destroy": [_vp], "ltx_vae_get_config": [_vp, _vp],
    "ltx_vae_set_noise_seed": [_vp, C.c_uint64], "ltx_vae_injects_noise": [_vp],
    "ltx_vae_latents_mean": [_vp], "ltx_vae_latents_std": [_vp],
    "ltx_vae_decode": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp],
    "ltx_vae_decode_tokens": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp],
    "ltx_vae_prepare_latents": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "ltx_guidance_step": [_vp, _vp, _vp, _i, _vp, _vp, _i, _i64, _f, _f, _f, _f, _vp, _vp],
    "ltx_guidance_step_stochastic": [_vp, _vp, _vp, _i, _vp, _vp, _i, _i64, _f, _f, _f, _f, _f, _vp, _vp, _vp],
    "ltx_sched_set_timesteps": [_vp, _i, _f, _i, _f, _f, _i, _vp, _vp],
    "ltx_sched_set_timesteps_ex": [_vp, _i, _f, _i, _f, _f, _i, _i, _i, _vp, _vp],
    "ltx_calculate_shift": [_i, _i, _i, _f, _f],
    "ltx_pcg32_randn": [C.c_uint64, C.c_uint64, _sz, _vp],
    "ltx_pcg32_u32": [C.c_uint64, C.c_uint64, _sz, _vp],
    "ltx_device_alloc": [_sz, _i, _vp], "ltx_device_free": [_vp], "ltx_memcpy_h2d": [_vp, _vp, _sz, _vp], "ltx_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "ltx_stream_synchronize": [_vp],
    "ltx_warmup": [_vp, _vp, _i, _i, _i, _i, _i, _vp], "ltx_set_autotune": [_i], "ltx_plan_save": [C.c_char_p], "ltx_plan_load": [C.c_char_p],
    "ltx_pipeline_last_steps": [C.POINTER(C.c_int), C.POINTER(C.c_int)], "ltx_attention_fallback_counts": [C.POINTER(C.c_ulonglong), C.c_int], "ltx_set_option": [C.c_char_p, C.c_char_p], "ltx_get_option": [C.c_char_p, C.c_char_p, C.c_int], "ltx_reset_options": [], "ltx_has_experiments": [],
    "ltx_build_video_coords": [_i, _i, _i, _i, _i, _i, _vp],
    "ltx_pipeline_params_default": [_vp],
    "ltx_pipeline_call": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp],
    "ltx_pipeline_last_timing": [_vp],
    "ltx_prof_enable": [_i], "ltx_prof_report": [_i, _vp, _vp, _vp], "ltx_prof_report_kernel": [_i, _i, _vp, _vp, _vp],
    "ltx_op_linear": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "ltx_op_linear_segmented": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ltx_op_ring_packed_bytes": [_i, _i], "ltx_op_ring_pack": [_vp, _i, _i, _vp, _vp],
    "ltx_op_linear_packed": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "ltx_op_rownorm_presum": [_vp, _vp, _i64, _i, _f, _vp, _vp, _vp, _i64, _i, _i, _vp, _i, _i, _vp],
    "ltx_op_linear_rowsq": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp], "ltx_op_rowsq": [_vp, _i64, _i, _i, _vp, _i, _vp],
    "ltx_op_linear_split_factor": [_i, _i, _i], "ltx_op_linear_deferred": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "ltx_op_rownorm_deferred": [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i64, _i, _i, _f, _vp, _vp, _i64, _i, _i, _vp],
    "ltx_op_attention_compact": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _i, _f, _vp, _vp],
    "ltx_op_attention_rowsq": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _i, _f, _vp],
    "ltx_op_rownorm": [_vp, _vp, _i64, _i, _i, _f, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "ltx_op_qknorm_rope": [_vp, _i64, _i, _i, _vp, _f, _vp, _vp, _i, _vp],
    "ltx_op_rope_table": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "ltx_op_timestep_embedding": [_vp, _i, _i, _f, _i, _vp, _vp],
    "ltx_op_attention": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp],
    "ltx_op_attention_prescaled": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ltx_op_conv3d": [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ltx_op_upsample3d": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ltx_op_conv_out_unpatchify": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ltx_op_blend": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ltx_op_gemm_plan": [_i, _i, _i, _i, _i, _i, _i, _i, C.c_char_p, _i],
    # include/ltxhip_t5.h
    "ltx_t5_config_default": [_vp], "ltx_t5_create": [_vp, _vp, _sz, _i, _i, _vp], "ltx_t5_destroy": [_vp],
    "ltx_t5_forward": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "ltx_t5_create_from_gguf": [_vp, C.c_char_p, _i, _i, _vp], "ltx_t5_forward_masked": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "ltx_gguf_open": [C.c_char_p, _vp], "ltx_gguf_close": [_vp], "ltx_gguf_count": [_vp], "ltx_gguf_find": [_vp, C.c_char_p],
    "ltx_gguf_tensor": [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp], "ltx_gguf_type_info": [_i, _vp, _vp],
    "ltx_gguf_dequantize": [_i, _vp, _i, _i64, _i, _vp, _vp],
    # include/ltxhip_frames.h
    "ltx_video_to_rgb8": [_vp, _i, _i, _i, _i, _vp, _vp], "ltx_write_png": [C.c_char_p, _vp, _i, _i],
    "ltx_save_frames_png": [_vp, _i, _i, _i, _i, C.c_char_p, _vp, _vp],
    "ltx_write_gif": [C.c_char_p, _vp, _i, _i, _i, _i, _i], "ltx_save_video_gif": [_vp, _i, _i, _i, _i, C.c_char_p, _vp],
    # include/ltxhip_presets.h
    "ltx_preset_count": [], "ltx_preset_name": [_i], "ltx_preset_get": [C.c_char_p, _vp], "ltx_pipeline_params_from_preset": [_vp, _vp],
    # include/ltxhip_weights.h
    "ltx_weights_detect_format": [C.c_char_p], "ltx_weights_remap_key": [C.c_char_p, C.c_char_p, _sz],
    "ltx_weights_is_transformer_key": [C.c_char_p], "ltx_weights_is_vae_key": [C.c_char_p],
    "ltx_name_mapper_create": [], "ltx_name_mapper_destroy": [_vp], "ltx_name_mapper_add": [_vp, _i, C.c_char_p, C.c_char_p],
    "ltx_name_mapper_has_mapping": [_vp, C.c_char_p], "ltx_name_mapper_map": [_vp, C.c_char_p, C.c_char_p, _sz],
    "ltx_weights_validate_names": [_vp, _sz, _vp, _sz, _vp, _vp],
    "ltx_safetensors_open": [C.c_char_p, _vp], "ltx_safetensors_close": [_vp], "ltx_safetensors_count": [_vp],
    "ltx_safetensors_tensor": [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp],
    "ltx_weights_resolve": [C.c_char_p, C.c_char_p, _sz, _vp],
    "ltx_dit_create_from_files": [_vp, C.c_char_p, _i, _i, _i, _vp], "ltx_vae_create_from_files": [_vp, C.c_char_p, _i, _i, _i, _vp],
    "ltx_vae_config_from_json": [C.c_char_p, _vp],
    # ltxhip_team.h: RCCL behind the C ABI (dlopen'ed on first use)
    "ltx_team_unique_id": [_vp], "ltx_team_create": [_vp, _i, _i, _i, _vp], "ltx_team_destroy": [_vp], "ltx_team_size": [_vp], "ltx_team_rank": [_vp],
    "ltx_team_allgather_f32": [_vp, _vp, _vp, _sz, _vp], "ltx_team_exchange_f32": [_vp, _vp, _sz, _i, _vp, _sz, _i, _vp],
}
EXPORTED_SYMBOLS = sorted(list(_SIGS) + ["ltx_last_error"])
for _name, _sig in _SIGS.items():
    _fn = getattr(lib, _name)          # raises AttributeError if the library lacks a declared symbol
    _fn.argtypes = _sig
    if _name not in ("ltx_calculate_shift", "ltx_vae_latents_mean", "ltx_vae_latents_std", "ltx_name_mapper_create",
                     "ltx_name_mapper_destroy", "ltx_safetensors_close", "ltx_safetensors_count", "ltx_team_destroy"):
        _fn.restype = C.c_int
lib.ltx_calculate_shift.restype = C.c_float
lib.ltx_op_ring_packed_bytes.restype = C.c_int64
lib.ltx_vae_latents_mean.restype = C.c_void_p
lib.ltx_vae_latents_std.restype = C.c_void_p
lib.ltx_name_mapper_create.restype = C.c_void_p
lib.ltx_name_mapper_destroy.restype = None
lib.ltx_safetensors_close.restype = None
lib.ltx_safetensors_count.restype = C.c_size_t
lib.ltx_gguf_close.restype = None
lib.ltx_gguf_count.restype = C.c_size_t


def _dt(t: torch.dtype) -> int:
    if t == torch.float32:
        return LTX_F32
    if t == torch.bfloat16:
        return LTX_BF16
    raise LtxError(f"unsupported dtype {t}")


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: torch.Tensor, dtype=None) -> torch.Tensor:
    if not t.is_cuda:
        raise LtxError("tensor must live on the GPU (the library takes device pointers)")
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _expect(name: str, t: torch.Tensor, shape):
    if tuple(t.shape) != tuple(shape):
        raise LtxError(f"{name}: shape {tuple(t.shape)} does not match the call's geometry {tuple(shape)}")


def _make_weights(weights: Dict[str, torch.Tensor]):
    arr = (_Weight * len(weights))()
    keep = []
    for i, (name, t) in enumerate(weights.items()):
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        t = t.contiguous()
        keep.append(t)
        nb = name.encode()
        keep.append(nb)
        arr[i].name = nb
        arr[i].data = t.data_ptr()
        arr[i].dtype = _dt(t.dtype)
        arr[i].ndim = t.dim()
        for j in range(t.dim()):
            arr[i].shape[j] = t.shape[j]
        arr[i].on_device = 1 if t.is_cuda else 0
    return arr, keep


def _floats(vals) -> "C.Array":
    return (C.c_float * len(vals))(*[float(v) for v in vals])


# ------------------------------------------------------------------ DiT
@dataclass
class LtxVideoTransformer3DModelConfig:          # ltx_transformer.rs:23-58
    in_channels: int = 128
    out_channels: int = 128
    patch_size: int = 1
    patch_size_t: int = 1
    num_attention_heads: int = 32
    attention_head_dim: int = 64
    cross_attention_dim: int = 2048
    num_layers: int = 28
    norm_eps: float = 1e-6
    caption_channels: int = 4096


class LtxVideoTransformer3DModel:
    """impl VideoTransformer3D (t2v_pipeline.rs:63-83) over ltx_dit_*."""

    def __init__(self, config: LtxVideoTransformer3DModelConfig, weights: Dict[str, torch.Tensor],
                 dtype: torch.dtype = torch.bfloat16, device: int = 0):
        self.config = config
        self.dtype = dtype
        c = DitConfigC(config.in_channels, config.out_channels, config.patch_size, config.patch_size_t,
                       config.num_attention_heads, config.attention_head_dim, config.cross_attention_dim,
                       config.num_layers, config.norm_eps, config.caption_channels)
        arr, keep = _make_weights(weights)
        self._h = C.c_void_p()
        _check(lib.ltx_dit_create(C.byref(c), arr, C.c_size_t(len(weights)), _dt(dtype), device, C.byref(self._h)))
        del keep

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib.ltx_dit_destroy(h)
            self._h = None

    @classmethod
    def from_files(cls, config: "LtxVideoTransformer3DModelConfig", path: str, unified: bool = False,
                   dtype: torch.dtype = torch.bfloat16, device: int = 0) -> "LtxVideoTransformer3DModel":
        """Build from a checkpoint on disk (ltx_dit_create_from_files): `unified` = Official single-file checkpoint
        (keys remapped/split as examples/ltx-video/main.rs:461-499), else the transformer's own file or directory."""
        self = cls.__new__(cls)
        self.config, self.dtype = config, dtype
        c = DitConfigC(config.in_channels, config.out_channels, config.patch_size, config.patch_size_t,
                       config.num_attention_heads, config.attention_head_dim, config.cross_attention_dim,
                       config.num_layers, config.norm_eps, config.caption_channels)
        self._h = C.c_void_p()
        _check(lib.ltx_dit_create_from_files(C.byref(c), path.encode(), int(unified), _dt(dtype), device, C.byref(self._h)))
        return self

    def context_cache(self, enable: bool):
        """keep the text-side projections (caption projection, cross-attention K/V) of repeated forwards with the same
        embeddings/mask pointers; LtxPipeline::call scopes it to one denoise loop."""
        _check(lib.ltx_dit_context_cache(self._h, int(enable)))
        self._ctx_cache = bool(enable)

    def set_skip_block_list(self, blocks: Sequence[int]):
        arr = (C.c_int * max(len(blocks), 1))(*blocks)
        _check(lib.ltx_dit_set_skip_blocks(self._h, arr, len(blocks)))

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, timestep,
                encoder_attention_mask: Optional[torch.Tensor], num_frames: int, height: int, width: int,
                rope_interpolation_scale: Optional[Tuple[float, float, float]] = None,
                video_coords: Optional[torch.Tensor] = None,
                skip_layer_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        io = hidden_states.dtype if hidden_states.dtype in (torch.float32, torch.bfloat16) else torch.float32
        h = _dev(hidden_states, io)
        e = _dev(encoder_hidden_states, io)
        if h.dim() != 3 or e.dim() != 3:
            raise LtxError("hidden_states must be [B,S,C] and encoder_hidden_states [B,K,C]")
        B, S, _ = h.shape
        K = e.shape[1]
        t = timestep.detach().float().flatten().cpu().tolist() if torch.is_tensor(timestep) else list(timestep)
        if len(t) != B:
            raise LtxError(f"timestep must have {B} entries")
        m = _dev(encoder_attention_mask, torch.float32) if encoder_attention_mask is not None else None
        if getattr(self, "_ctx_cache", False):
            # the cache identifies a context by its device POINTERS: a converted / re-laid-out temporary is freed after this
            # call and its address reused by the next one, which would hit the cache with another context's K/V
            if e.data_ptr() != encoder_hidden_states.data_ptr() or (m is not None and m.data_ptr() != encoder_attention_mask.data_ptr()):
                raise LtxError("context_cache(True) needs encoder_hidden_states / encoder_attention_mask that are already contiguous and of "
                               "the dtype the call uses (hidden_states' dtype / float32): a temporary copy has no stable identity")
        vc = _dev(video_coords, torch.float32) if video_coords is not None else None
        if getattr(self, "_ctx_cache", False) and vc is not None and vc.data_ptr() != video_coords.data_ptr():
            raise LtxError("context_cache(True) needs video_coords that are already a contiguous float32 device tensor: the RoPE "
                           "tables of the scope are kept per coords POINTER, a temporary copy has no stable identity")
        slm = None
        if skip_layer_mask is not None:
            slm = _floats(skip_layer_mask.detach().float().cpu().flatten().tolist())
        rs = _floats(rope_interpolation_scale) if rope_interpolation_scale is not None else None
        out = torch.empty(B, S, self.config.out_channels, dtype=io, device=h.device)
        _check(lib.ltx_dit_forward(self._h, _ptr(h), _ptr(e), _floats(t), _ptr(m), B, S, K, num_frames, height, width,
                                   rs, _ptr(vc), slm, _dt(io), _ptr(out), _stream()))
        return out


# ------------------------------------------------------------------ T5 text encoder (include/ltxhip_t5.h)
class T5ConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("d_model", C.c_int), ("d_kv", C.c_int), ("d_ff", C.c_int), ("num_layers", C.c_int),
                ("num_heads", C.c_int), ("relative_attention_num_buckets", C.c_int), ("relative_attention_max_distance", C.c_int),
                ("layer_norm_epsilon", C.c_float)]


@dataclass
class T5EncoderConfig:                           # text_encoder.rs:66-113; defaults = t5_xxl() (:169-184)
    vocab_size: int = 32128
    d_model: int = 4096
    d_kv: int = 64
    d_ff: int = 10240
    num_layers: int = 24
    num_heads: int = 64
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6


class T5TextEncoder:
    """impl VTextEncoder (text_encoder.rs:597-606) over ltx_t5_*: forward(input_ids) -> [B,S,d_model] in the model dtype."""

    def __init__(self, config: T5EncoderConfig, weights: Optional[Dict[str, torch.Tensor]], dtype: torch.dtype = torch.bfloat16, device: int = 0,
                 gguf_path: Optional[str] = None):
        self.config, self.dtype = config, dtype
        c = T5ConfigC(config.vocab_size, config.d_model, config.d_kv, config.d_ff, config.num_layers, config.num_heads,
                      config.relative_attention_num_buckets, config.relative_attention_max_distance, config.layer_norm_epsilon)
        self._h = C.c_void_p()
        if gguf_path is not None:
            _check(lib.ltx_t5_create_from_gguf(C.byref(c), os.fsencode(gguf_path), _dt(dtype), device, C.byref(self._h)))
            return
        arr, keep = _make_weights(weights)
        _check(lib.ltx_t5_create(C.byref(c), arr, C.c_size_t(len(weights)), _dt(dtype), device, C.byref(self._h)))
        del keep

    @classmethod
    def from_gguf(cls, gguf_path: str, config: Optional[T5EncoderConfig] = None, dtype: torch.dtype = torch.float32, device: int = 0):
        """QuantizedT5EncoderModel::load_with_config (quantized_t5_encoder.rs:575-603): the reference's default text encoder"""
        return cls(config or T5EncoderConfig(), None, dtype, device, gguf_path=gguf_path)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:                    # `lib` is already None when the interpreter is shutting down
            lib.ltx_t5_destroy(h)
            self._h = None

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """forward(input_ids) (text_encoder.rs:600-605) / forward(input_ids, Some(mask)) (quantized_t5_encoder.rs:608-650)"""
        ids = input_ids.detach().to("cpu", torch.int32).contiguous()
        if ids.dim() != 2:
            raise LtxError("input_ids must be [B, S]")
        B, S = ids.shape
        out = torch.empty(B, S, self.config.d_model, dtype=self.dtype, device=torch.device("cuda", torch.cuda.current_device()))
        if attention_mask is not None:
            am = attention_mask.detach().to("cpu", torch.float32).contiguous()
            if tuple(am.shape) != (B, S):
                raise LtxError("attention_mask must be [B, S]")
            _check(lib.ltx_t5_forward_masked(self._h, C.c_void_p(ids.data_ptr()), C.c_void_p(am.data_ptr()), B, S, _dt(self.dtype), _ptr(out), _stream()))
        else:
            _check(lib.ltx_t5_forward(self._h, C.c_void_p(ids.data_ptr()), B, S, _dt(self.dtype), _ptr(out), _stream()))
        torch.cuda.current_stream().synchronize()       # the ids were read from host memory asynchronously
        return out


# ------------------------------------------------------------------ GGUF (include/ltxhip_weights.h)
GGML_TYPES = {"F32": 0, "F16": 1, "Q4_0": 2, "Q5_0": 6, "Q8_0": 8, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14, "BF16": 30}


def gguf_type_info(ggml_type: int):
    be, bb = C.c_int(), C.c_int()
    _check(lib.ltx_gguf_type_info(ggml_type, C.byref(be), C.byref(bb)))
    return be.value, bb.value


def gguf_tensors(path: str):
    """[(name, ggml_type, shape outermost-first, raw bytes)] of a GGUF file, in file order (host-side parser, no GPU needed)"""
    h = C.c_void_p()
    _check(lib.ltx_gguf_open(os.fsencode(path), C.byref(h)))
    try:
        out = []
        for i in range(lib.ltx_gguf_count(h)):
            nm, ty, nd, shp, data, nb = C.c_char_p(), C.c_int(), C.c_int(), C.POINTER(C.c_int64)(), C.c_void_p(), C.c_size_t()
            rc = lib.ltx_gguf_tensor(h, i, C.byref(nm), C.byref(ty), C.byref(nd), C.byref(shp), C.byref(data), C.byref(nb))
            raw = C.string_at(data, nb.value) if rc == 0 else None
            out.append((nm.value.decode(), ty.value, tuple(shp[k] for k in range(nd.value)), raw))
        return out
    finally:
        lib.ltx_gguf_close(h)


def gguf_dequantize(ggml_type: int, blocks: bytes, numel: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """QTensor::dequantize on the device: raw ggml blocks (host bytes) -> dense [numel] tensor"""
    out = torch.empty(numel, dtype=dtype, device=torch.device("cuda", torch.cuda.current_device()))
    buf = C.create_string_buffer(blocks, len(blocks))
    _check(lib.ltx_gguf_dequantize(ggml_type, buf, 0, numel, _dt(dtype), _ptr(out), _stream()))
    torch.cuda.current_stream().synchronize()
    return out


# ------------------------------------------------------------------ VAE
@dataclass
class AutoencoderKLLtxVideoConfig:               # vae.rs:32-103 (decoder side)
    latent_channels: int = 128
    out_channels: int = 3
    decoder_block_out_channels: Tuple[int, ...] = (256, 512, 1024)
    decoder_layers_per_block: Tuple[int, ...] = (5, 5, 5, 5)
    decoder_upsample_factor: Tuple[int, ...] = (2, 2, 2)
    patch_size: int = 4
    patch_size_t: int = 1
    timestep_conditioning: bool = True
    decoder_causal: bool = False
    scaling_factor: float = 1.0
    spatial_compression_ratio: int = 32
    temporal_compression_ratio: int = 8
    decoder_inject_noise: Tuple[bool, ...] = (False, False, False, False)     # vae.rs:87; blocks whose resnets add noise[h, w] * per_channel_scale (set_noise_seed)
    decoder_upsample_residual: Tuple[bool, ...] = (True, True, True)            # vae.rs:88
    decoder_spatiotemporal_scaling: Tuple[bool, ...] = (True, True, True)       # vae.rs:78; False: that up-block's upsampler is the spatial-only (1, 2, 2) form
    resnet_eps: float = 1e-6                                                    # vae.rs:83 (norm3 only: unused by the decoder)


class AutoencoderKLLtxVideo:
    """impl VaeLtxVideo (t2v_pipeline.rs:91-103) over ltx_vae_*; tiling fields as vae.rs:1744-1758."""

    @staticmethod
    def _config_c(config: "AutoencoderKLLtxVideoConfig") -> "VaeConfigC":
        c = VaeConfigC()
        c.latent_channels, c.out_channels = config.latent_channels, config.out_channels
        nb = len(config.decoder_block_out_channels)
        c.n_blocks = nb
        for i in range(nb):
            c.decoder_block_out_channels[i] = config.decoder_block_out_channels[i]
            c.decoder_upsample_factor[i] = config.decoder_upsample_factor[i]
        for i in range(nb + 1):
            c.decoder_layers_per_block[i] = config.decoder_layers_per_block[i]
        for i, x in enumerate(config.decoder_inject_noise[:5]): c.decoder_inject_noise[i] = int(x)
        for i, x in enumerate(config.decoder_upsample_residual[:4]): c.decoder_upsample_residual[i] = int(x)
        for i, x in enumerate(config.decoder_spatiotemporal_scaling[:4]): c.decoder_spatiotemporal_scaling[i] = int(x)
        c.resnet_eps = config.resnet_eps
        c.patch_size, c.patch_size_t = config.patch_size, config.patch_size_t
        c.timestep_conditioning = int(config.timestep_conditioning)
        c.decoder_causal = int(config.decoder_causal)
        c.scaling_factor = config.scaling_factor
        c.spatial_compression_ratio = config.spatial_compression_ratio
        c.temporal_compression_ratio = config.temporal_compression_ratio
        return c

    def _tiling_defaults(self):
        # defaults of examples/ltx-video/main.rs:514-516 (tiling off unless asked)
        self.use_tiling = False
        self.use_framewise_decoding = False
        self.tile_sample_min_height = 512
        self.tile_sample_min_width = 512
        self.tile_sample_min_num_frames = 16
        self.tile_sample_stride_height = 384
        self.tile_sample_stride_width = 384
        self.tile_sample_stride_num_frames = 8

    def __init__(self, config: AutoencoderKLLtxVideoConfig, weights: Dict[str, torch.Tensor],
                 dtype: torch.dtype = torch.bfloat16, device: int = 0):
        self.config = config
        self.dtype = dtype
        c = self._config_c(config)
        arr, keep = _make_weights(weights)
        self._h = C.c_void_p()
        _check(lib.ltx_vae_create(C.byref(c), arr, C.c_size_t(len(weights)), _dt(dtype), device, C.byref(self._h)))
        del keep
        self._tiling_defaults()

    @classmethod
    def from_files(cls, config: "AutoencoderKLLtxVideoConfig", path: str, unified: bool = False,
                   dtype: torch.dtype = torch.bfloat16, device: int = 0) -> "AutoencoderKLLtxVideo":
        """Build from a checkpoint on disk (ltx_vae_create_from_files); see LtxVideoTransformer3DModel.from_files."""
        self = cls.__new__(cls)
        self.config, self.dtype = config, dtype
        c = cls._config_c(config)
        self._h = C.c_void_p()
        _check(lib.ltx_vae_create_from_files(C.byref(c), path.encode(), int(unified), _dt(dtype), device, C.byref(self._h)))
        self.config = self._config_py(self.get_config())       # a vae/config.json beside the weights replaces `config` (main.rs:525-534)
        self._tiling_defaults()
        return self

    def set_noise_seed(self, seed: int) -> None:
        """Noise injection (decoder_inject_noise, vae.rs:741-753): plane k of this handle's life is Pcg32(seed, k).randn(H * W);
        restarts k at 0 (ltx_vae_set_noise_seed)."""
        _check(lib.ltx_vae_set_noise_seed(self._h, C.c_uint64(seed)))

    def injects_noise(self) -> bool:
        """whether any resnet of this decoder injects noise (the flag AND a per_channel_scaleN.weight in the checkpoint, vae.rs:676-689)"""
        return bool(lib.ltx_vae_injects_noise(self._h))

    def get_config(self) -> "VaeConfigC":
        """the engine's effective config (ltx_vae_get_config)"""
        c = VaeConfigC()
        _check(lib.ltx_vae_get_config(self._h, C.byref(c)))
        return c

    @staticmethod
    def _config_py(c: "VaeConfigC") -> "AutoencoderKLLtxVideoConfig":
        nb = c.n_blocks
        return AutoencoderKLLtxVideoConfig(
            latent_channels=c.latent_channels, out_channels=c.out_channels,
            decoder_block_out_channels=tuple(c.decoder_block_out_channels[:nb]), decoder_layers_per_block=tuple(c.decoder_layers_per_block[:nb + 1]),
            decoder_upsample_factor=tuple(c.decoder_upsample_factor[:nb]), patch_size=c.patch_size, patch_size_t=c.patch_size_t,
            timestep_conditioning=bool(c.timestep_conditioning), decoder_causal=bool(c.decoder_causal), scaling_factor=c.scaling_factor,
            spatial_compression_ratio=c.spatial_compression_ratio, temporal_compression_ratio=c.temporal_compression_ratio,
            decoder_inject_noise=tuple(bool(x) for x in c.decoder_inject_noise[:nb + 1]),
            decoder_upsample_residual=tuple(bool(x) for x in c.decoder_upsample_residual[:nb]),
            decoder_spatiotemporal_scaling=tuple(bool(x) for x in c.decoder_spatiotemporal_scaling[:nb]), resnet_eps=c.resnet_eps)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib.ltx_vae_destroy(h)
            self._h = None

    def _tiling(self) -> Optional[TilingC]:
        if not (self.use_tiling or self.use_framewise_decoding):
            return None
        return TilingC(int(self.use_tiling), int(self.use_framewise_decoding), self.tile_sample_min_height,
                       self.tile_sample_min_width, self.tile_sample_min_num_frames, self.tile_sample_stride_height,
                       self.tile_sample_stride_width, self.tile_sample_stride_num_frames)

    def decode(self, latents: torch.Tensor, timestep=None, postprocess: bool = False, rgb8: bool = False) -> torch.Tensor:
        """rgb8=True: the post-processed video as u8 frames [B, frames, H, W, 3] (main.rs:653-675) from the decoder's last epilogue"""
        io = latents.dtype if latents.dtype in (torch.float32, torch.bfloat16) else torch.float32
        z = _dev(latents, io)
        if z.dim() != 5 or z.shape[1] != self.config.latent_channels:
            raise LtxError("latents must be [B, latent_channels, F, H, W]")
        B, _, F, H, W = z.shape
        t = None
        if timestep is not None:
            tv = timestep.detach().float().flatten().cpu().tolist() if torch.is_tensor(timestep) else list(timestep)
            if len(tv) != B:
                raise LtxError(f"timestep must have {B} entries")
            t = _floats(tv)
        r, tr = self.config.spatial_compression_ratio, self.config.temporal_compression_ratio
        if rgb8:
            out = torch.empty(B, (F - 1) * tr + 1, H * r, W * r, self.config.out_channels, dtype=torch.uint8, device=z.device)
        else:
            out = torch.empty(B, self.config.out_channels, (F - 1) * tr + 1, H * r, W * r, dtype=torch.float32, device=z.device)
        tl = self._tiling()
        _check(lib.ltx_vae_decode(self._h, _ptr(z), _dt(io), t, B, F, H, W, C.byref(tl) if tl else None, 2 if rgb8 else int(postprocess),
                                  _ptr(out), _stream()))
        return out

    def decode_tokens(self, tokens: torch.Tensor, F: int, H: int, W: int, timestep=None, noise: Optional[torch.Tensor] = None,
                      noise_scale=None, postprocess: bool = False) -> torch.Tensor:
        x = _dev(tokens, torch.float32)
        B = x.shape[0]
        t = _floats(list(timestep)) if timestep is not None else None
        ns = _floats(list(noise_scale)) if noise_scale is not None else None
        nz = _dev(noise, torch.float32) if noise is not None else None
        r, tr = self.config.spatial_compression_ratio, self.config.temporal_compression_ratio
        out = torch.empty(B, self.config.out_channels, (F - 1) * tr + 1, H * r, W * r, dtype=torch.float32, device=x.device)
        tl = self._tiling()
        _check(lib.ltx_vae_decode_tokens(self._h, _ptr(x), _ptr(nz), ns, t, B, F, H, W, C.byref(tl) if tl else None,
                                         int(postprocess), _ptr(out), _stream()))
        return out

    def prepare_latents(self, tokens: torch.Tensor, F: int, H: int, W: int, noise: Optional[torch.Tensor] = None,
                        noise_scale=None) -> torch.Tensor:
        """denormalize (+ decode-noise mix) of packed tokens [B,S,C] -> [B,C,F,H,W] f32 (t2v_pipeline.rs:1002-1053)."""
        x = _dev(tokens, torch.float32)
        B = x.shape[0]
        ns = _floats(list(noise_scale)) if noise_scale is not None else None
        nz = _dev(noise, torch.float32) if noise is not None else None
        out = torch.empty_like(x)
        _check(lib.ltx_vae_prepare_latents(self._h, _ptr(x), _ptr(nz), ns, B, F, H, W, _ptr(out), _stream()))
        return out.reshape(B, F, H, W, -1).permute(0, 4, 1, 2, 3).contiguous()

    def latents_mean(self) -> torch.Tensor:
        lib.ltx_vae_latents_mean.restype = C.c_void_p
        p = lib.ltx_vae_latents_mean(self._h)
        out = torch.empty(self.config.latent_channels, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        _hip_memcpy_d2d(out.data_ptr(), p, 4 * self.config.latent_channels)
        return out

    def latents_std(self) -> torch.Tensor:
        lib.ltx_vae_latents_std.restype = C.c_void_p
        p = lib.ltx_vae_latents_std(self._h)
        out = torch.empty(self.config.latent_channels, dtype=torch.float32, device="cuda")
        _hip_memcpy_d2d(out.data_ptr(), p, 4 * self.config.latent_channels)
        return out


def _hip_memcpy_d2d(dst: int, src: int, nbytes: int):
    hip = C.CDLL("libamdhip64.so")
    rc = hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3)  # hipMemcpyDeviceToDevice
    if rc != 0:
        raise LtxError(f"hipMemcpy failed rc={rc}")


# ------------------------------------------------------------------ scheduler / host helpers
lib.ltx_calculate_shift.restype = C.c_float


def calculate_shift(seq_len: int, base_seq_len: int = 256, max_seq_len: int = 4096,
                    base_shift: float = 0.5, max_shift: float = 1.15) -> float:
    return float(lib.ltx_calculate_shift(seq_len, base_seq_len, max_seq_len, C.c_float(base_shift), C.c_float(max_shift)))


class FlowMatchEulerDiscreteScheduler:
    """Scheduler trait (t2v_pipeline.rs:28-37) as implemented at scheduler.rs:646-668."""

    def __init__(self, shift: float = 1.0, shift_terminal: Optional[float] = 0.1, stochastic_sampling: bool = False,
                 use_karras_sigmas: bool = False, use_exponential_sigmas: bool = False, use_beta_sigmas: bool = False, invert_sigmas: bool = False):
        if int(use_karras_sigmas) + int(use_exponential_sigmas) + int(use_beta_sigmas) > 1:      # scheduler.rs:85-93
            raise LtxError("Only one of use_beta_sigmas/use_exponential_sigmas/use_karras_sigmas can be enabled.")
        self.shift = shift
        self.shift_terminal = shift_terminal
        self.stochastic_sampling = stochastic_sampling
        self.sigma_kind = 1 if use_karras_sigmas else (2 if use_exponential_sigmas else (3 if use_beta_sigmas else 0))
        self.invert_sigmas = invert_sigmas
        self.sigmas: List[float] = []
        self.timesteps: List[int] = []
        self.step_index = 0

    def set_timesteps(self, sigmas: Sequence[float], mu: Optional[float]) -> List[int]:
        n = len(sigmas)
        sin = _floats(sigmas)
        sout = (C.c_float * (n + 1))()
        tout = (C.c_int64 * n)()
        _check(lib.ltx_sched_set_timesteps_ex(sin, n, C.c_float(mu if mu is not None else 0.0), int(mu is not None),
                                              C.c_float(self.shift), C.c_float(self.shift_terminal or 0.0),
                                              int(self.shift_terminal is not None), self.sigma_kind, int(self.invert_sigmas), sout, tout))
        self.sigmas = list(sout)
        self.timesteps = list(tout)
        self.step_index = 0
        return self.timesteps

    def step(self, noise_pred: torch.Tensor, timestep: int, latents: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x + (sigma_next - sigma) * v, or with stochastic_sampling (1 - sigma_next) * (x - sigma * v) + sigma_next * noise
        (`noise` = the randn_like(sample) draw, supplied by the caller); f32, on a copy (scheduler.rs:544-581)."""
        sg, sn = C.c_float(self.sigmas[self.step_index]).value, C.c_float(self.sigmas[self.step_index + 1]).value
        dt = sn - sg
        x = _dev(latents, torch.float32).clone()
        p = _dev(noise_pred)
        B = x.shape[0]
        if self.stochastic_sampling:
            if noise is None:
                raise LtxError("stochastic_sampling: pass the per-step noise (the reference draws randn_like(sample))")
            nz = _dev(noise, torch.float32)
            _check(lib.ltx_guidance_step_stochastic(_ptr(p), None, None, _dt(p.dtype), _ptr(x), None, B, C.c_int64(x[0].numel()),
                                                    C.c_float(1.0), C.c_float(0.0), C.c_float(0.0), C.c_float(sg), C.c_float(sn),
                                                    _ptr(nz), None, _stream()))
            self.step_index += 1
            return x
        _check(lib.ltx_guidance_step(_ptr(p), None, None, _dt(p.dtype), _ptr(x), None, B, C.c_int64(x[0].numel()),
                                     C.c_float(1.0), C.c_float(0.0), C.c_float(0.0), C.c_float(dt), None, _stream()))
        self.step_index += 1
        return x


def guidance_combine(text, uncond=None, perturbed=None, guidance_scale=1.0, guidance_rescale=0.0, stg_scale=0.0):
    """CFG/STG mix of t2v_pipeline.rs:941-964 (returns the f32 combined prediction)."""
    t = _dev(text)
    u = _dev(uncond, t.dtype) if uncond is not None else None
    p = _dev(perturbed, t.dtype) if perturbed is not None else None
    B = t.shape[0]
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    ws = torch.zeros(8 * B, dtype=torch.float64, device=t.device)
    _check(lib.ltx_guidance_step(_ptr(t), _ptr(u), _ptr(p), _dt(t.dtype), None, _ptr(out), B, C.c_int64(t[0].numel()),
                                 C.c_float(guidance_scale), C.c_float(guidance_rescale), C.c_float(stg_scale), C.c_float(0.0),
                                 _ptr(ws), _stream()))
    return out


def pcg32_randn(seed: int, shape: Sequence[int], inc: int = 1442695040888963407) -> torch.Tensor:
    """Pcg32::new(seed, inc).randn(shape) (deterministic_rng.rs; main.rs:568) -> CPU f32 tensor."""
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=torch.float32)
    _check(lib.ltx_pcg32_randn(C.c_uint64(seed), C.c_uint64(inc), C.c_size_t(n), C.c_void_p(out.data_ptr())))
    return out.reshape(*shape)


def pcg32_u32(seed: int, n: int, inc: int = 1442695040888963407) -> torch.Tensor:
    """first n outputs of Pcg32::new(seed, inc).next_u32() as int64 (exact)"""
    import numpy as np
    out = np.empty(n, dtype=np.uint32)
    _check(lib.ltx_pcg32_u32(C.c_uint64(seed), C.c_uint64(inc), C.c_size_t(n), C.c_void_p(out.ctypes.data)))
    return torch.from_numpy(out.astype(np.int64))


def build_video_coords(F: int, H: int, W: int, frame_rate: int = 25, ts_ratio: int = 8, sp_ratio: int = 32) -> torch.Tensor:
    out = torch.empty(F * H * W, 3, dtype=torch.float32)
    _check(lib.ltx_build_video_coords(F, H, W, frame_rate, ts_ratio, sp_ratio, C.c_void_p(out.data_ptr())))
    return out


def pack_latents(x: torch.Tensor) -> torch.Tensor:
    """LtxPipeline::pack_latents with patch sizes 1 (t2v_pipeline.rs:474-504): a pure layout view."""
    b, c, f, h, w = x.shape
    return x.permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c).contiguous()


# ------------------------------------------------------------------ pipeline
@dataclass
class PipelineCall:
    height: int = 512
    width: int = 768
    num_frames: int = 97
    frame_rate: int = 25
    num_inference_steps: int = 7
    sigmas: Optional[List[float]] = None
    guidance_scale: float = 1.0
    guidance_rescale: float = 0.0
    stg_scale: float = 0.0
    skip_block_list: Optional[List[int]] = None
    decode_timestep: float = 0.05
    decode_noise_scale: float = 0.025
    output_latent: bool = False
    postprocess: bool = True
    output_rgb8: bool = False                   # the video as u8 frames [B, frames, H, W, 3] (ltx_pipeline_params::postprocess = 2; main.rs:653-675)
    shift_terminal: Optional[float] = 0.1
    stochastic_sampling: bool = False           # SchedulerConfig::stochastic_sampling (0.9.6-distilled preset)


class LtxPipeline:
    """LtxPipeline (t2v_pipeline.rs:245-302) holding the two boxed components that matter here."""

    def __init__(self, transformer: LtxVideoTransformer3DModel, vae: Optional[AutoencoderKLLtxVideo]):
        self.transformer = transformer
        self.vae = vae
        self.last_timing_ms = (0.0, 0.0, 0.0, 0.0)

    def call(self, args: PipelineCall, latents: torch.Tensor, prompt_embeds: torch.Tensor, prompt_attention_mask: torch.Tensor,
             negative_prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_attention_mask: Optional[torch.Tensor] = None,
             decode_noise: Optional[torch.Tensor] = None, step_noise: Optional[torch.Tensor] = None,
             interrupt=None, on_step=None):
        """Returns (final_latents [B,S,C] f32, video [B,3,frames,H,W] f32 or None).
        step_noise [steps,B,S,C] f32: the per-step draws of the stochastic-sampling scheduler (required iff enabled).
        interrupt: a ctypes.c_int the caller may set from another thread (LtxPipeline::interrupt, t2v_pipeline.rs:266, 861-863);
        on_step(step, num_steps, timestep) -> truthy to stop: the per-step hook.  self.last_steps = (executed, requested)."""
        lat = _dev(latents, torch.float32).clone()
        pe = _dev(prompt_embeds, torch.float32)
        pm = _dev(prompt_attention_mask, torch.float32)
        ne = _dev(negative_prompt_embeds, torch.float32) if negative_prompt_embeds is not None else None
        nm = _dev(negative_prompt_attention_mask, torch.float32) if negative_prompt_attention_mask is not None else None
        dn = _dev(decode_noise, torch.float32) if decode_noise is not None else None
        if pe.dim() != 3 or lat.dim() != 3:
            raise LtxError("latents must be [B, F*H*W, in_channels] and prompt_embeds [B, K, caption_channels]")
        B, K = pe.shape[0], pe.shape[1]
        # latent geometry as LtxPipeline::call derives it (t2v_pipeline.rs:743-747), checked against every tensor handed over:
        # the C entry takes raw pointers, a mismatch would be out-of-bounds device traffic
        tr = self.vae.config.temporal_compression_ratio if self.vae is not None else 8
        sr = self.vae.config.spatial_compression_ratio if self.vae is not None else 32
        F, H, W = (args.num_frames - 1) // tr + 1, args.height // sr, args.width // sr
        Cin = self.transformer.config.in_channels
        _expect("latents", lat, (B, F * H * W, Cin))
        _expect("prompt_embeds", pe, (B, K, self.transformer.config.caption_channels))
        _expect("prompt_attention_mask", pm, (B, K))
        if ne is not None: _expect("negative_prompt_embeds", ne, (B, K, self.transformer.config.caption_channels))
        if nm is not None: _expect("negative_prompt_attention_mask", nm, (B, K))
        if dn is not None: _expect("decode_noise", dn, (B, Cin, F, H, W))
        if step_noise is not None: _expect("step_noise", step_noise, (args.num_inference_steps, B, F * H * W, Cin))
        p = PipelineParamsC()
        lib.ltx_pipeline_params_default(C.byref(p))
        p.height, p.width, p.num_frames, p.frame_rate = args.height, args.width, args.num_frames, args.frame_rate
        p.num_inference_steps = args.num_inference_steps
        keep = []
        if args.sigmas is not None:
            s = _floats(args.sigmas); keep.append(s)
            p.sigmas = C.cast(s, C.POINTER(C.c_float))
        p.guidance_scale, p.guidance_rescale, p.stg_scale = args.guidance_scale, args.guidance_rescale, args.stg_scale
        if args.skip_block_list is not None:
            sb = (C.c_int * max(len(args.skip_block_list), 1))(*args.skip_block_list); keep.append(sb)
            p.skip_block_list = C.cast(sb, C.POINTER(C.c_int))
            p.n_skip_blocks = len(args.skip_block_list)
        p.decode_timestep, p.decode_noise_scale = args.decode_timestep, args.decode_noise_scale
        p.output_latent, p.postprocess = int(args.output_latent), (2 if args.output_rgb8 else int(args.postprocess))
        p.shift_terminal = args.shift_terminal if args.shift_terminal is not None else 0.0
        p.use_shift_terminal = int(args.shift_terminal is not None)
        if args.stochastic_sampling:
            if step_noise is None:
                raise LtxError("stochastic_sampling needs step_noise [steps,B,S,C]")
            sn = _dev(step_noise, torch.float32); keep.append(sn)
            p.stochastic_sampling = 1
            p.step_noise = C.cast(C.c_void_p(sn.data_ptr()), C.POINTER(C.c_float))
        tl = self.vae._tiling() if self.vae is not None else None
        if tl is not None:
            keep.append(tl)
            p.tiling = C.pointer(tl)
        if interrupt is not None:
            p.interrupt = C.pointer(interrupt)
        hook_exc = []
        if on_step is not None:
            def _hook(_user, step, num_steps, timestep):
                try:
                    return 1 if on_step(step, num_steps, timestep) else 0
                except BaseException as exc:                  # never unwind through the C frame: stop the loop, re-raise after the call
                    hook_exc.append(exc)
                    return 1
            cb = StepFn(_hook); keep.append(cb)
            p.on_step = C.cast(cb, C.c_void_p)
        video = None
        if not args.output_latent:
            if self.vae is None:
                raise LtxError("decode requested but the pipeline has no VAE")
            # the decoder's own output shape (vae.rs:2101-2136): (F-1)*tr+1 frames, which is num_frames only for tr*k+1
            if args.output_rgb8:
                video = torch.empty(B, (F - 1) * tr + 1, H * sr, W * sr, 3, dtype=torch.uint8, device=lat.device)
            else:
                video = torch.empty(B, 3, (F - 1) * tr + 1, H * sr, W * sr, dtype=torch.float32, device=lat.device)
        _check(lib.ltx_pipeline_call(self.transformer._h, self.vae._h if self.vae is not None else None, C.byref(p),
                                     _ptr(lat), _ptr(pe), _ptr(pm), _ptr(ne), _ptr(nm), _ptr(dn), B, K,
                                     _ptr(video), _stream()))
        ms = (C.c_float * 4)()
        _check(lib.ltx_pipeline_last_timing(ms))
        self.last_timing_ms = tuple(ms)
        ex, rq = C.c_int(0), C.c_int(0)
        _check(lib.ltx_pipeline_last_steps(C.byref(ex), C.byref(rq)))
        self.last_steps = (ex.value, rq.value)
        if hook_exc:
            raise hook_exc[0]
        return lat, video


# ------------------------------------------------------------------ version presets (include/ltxhip_presets.h; configs.rs:50-283)
class PresetC(C.Structure):
    _fields_ = [("version", C.c_char_p), ("guidance_scale", C.c_float), ("num_inference_steps", C.c_int), ("stg_scale", C.c_float),
                ("rescaling_scale", C.c_float), ("stochastic_sampling", C.c_int), ("skip_block_list", C.c_int * 8), ("n_skip_blocks", C.c_int),
                ("timesteps", C.c_float * 16), ("n_timesteps", C.c_int), ("decode_timestep", C.c_float), ("has_decode_timestep", C.c_int),
                ("decode_noise_scale", C.c_float), ("has_decode_noise_scale", C.c_int), ("transformer", DitConfigC), ("vae", VaeConfigC),
                ("vae_encoder_block_out_channels", C.c_int * 5), ("vae_encoder_layers_per_block", C.c_int * 5),
                ("num_train_timesteps", C.c_int), ("shift", C.c_float), ("use_dynamic_shifting", C.c_int), ("base_shift", C.c_float),
                ("max_shift", C.c_float), ("base_image_seq_len", C.c_int), ("max_image_seq_len", C.c_int), ("shift_terminal", C.c_float),
                ("has_shift_terminal", C.c_int), ("time_shift_exponential", C.c_int)]


@dataclass
class LTXVFullConfig:
    """get_config_by_version's result (configs.rs:40-46): inference settings + the three component configs."""
    version: str
    guidance_scale: float
    num_inference_steps: int
    stg_scale: float
    rescaling_scale: float
    stochastic_sampling: bool
    skip_block_list: List[int]
    timesteps: Optional[List[float]]
    decode_timestep: Optional[float]
    decode_noise_scale: Optional[float]
    transformer: "LtxVideoTransformer3DModelConfig"
    vae: "AutoencoderKLLtxVideoConfig"
    vae_encoder_block_out_channels: List[int]
    vae_encoder_layers_per_block: List[int]
    scheduler: Dict[str, object]

    def pipeline_call(self, height: int, width: int, num_frames: int, **over) -> "PipelineCall":
        """the PipelineCall examples/ltx-video/main.rs:585-646 builds from this preset (ltx_pipeline_params_from_preset)"""
        kw = dict(height=height, width=width, num_frames=num_frames,
                  num_inference_steps=len(self.timesteps) if self.timesteps else self.num_inference_steps, sigmas=self.timesteps,
                  guidance_scale=self.guidance_scale, guidance_rescale=self.rescaling_scale, stg_scale=self.stg_scale,
                  skip_block_list=list(self.skip_block_list) or None,
                  decode_timestep=self.decode_timestep if self.decode_timestep is not None else 0.0,
                  stochastic_sampling=self.stochastic_sampling)
        kw["decode_noise_scale"] = self.decode_noise_scale if self.decode_noise_scale is not None else kw["decode_timestep"]
        kw.update(over)
        return PipelineCall(**kw)


def preset_names() -> List[str]:
    lib.ltx_preset_name.restype = C.c_char_p
    return [lib.ltx_preset_name(i).decode() for i in range(lib.ltx_preset_count())]


def get_config_by_version(version: str) -> LTXVFullConfig:
    """configs.rs:50-70 (aliases; unknown strings fall back to 0.9.5)."""
    p = PresetC()
    _check(lib.ltx_preset_get(version.encode(), C.byref(p)))
    t, v = p.transformer, p.vae
    tcfg = LtxVideoTransformer3DModelConfig(in_channels=t.in_channels, out_channels=t.out_channels, patch_size=t.patch_size, patch_size_t=t.patch_size_t,
                                            num_attention_heads=t.num_attention_heads, attention_head_dim=t.attention_head_dim,
                                            cross_attention_dim=t.cross_attention_dim, num_layers=t.num_layers, norm_eps=t.norm_eps,
                                            caption_channels=t.caption_channels)
    nb = v.n_blocks
    vcfg = AutoencoderKLLtxVideoConfig(latent_channels=v.latent_channels, out_channels=v.out_channels,
                                       decoder_block_out_channels=tuple(v.decoder_block_out_channels[:nb]),
                                       decoder_layers_per_block=tuple(v.decoder_layers_per_block[:nb + 1]),
                                       decoder_upsample_factor=tuple(v.decoder_upsample_factor[:nb]), patch_size=v.patch_size,
                                       patch_size_t=v.patch_size_t, timestep_conditioning=bool(v.timestep_conditioning),
                                       decoder_causal=bool(v.decoder_causal), scaling_factor=v.scaling_factor,
                                       spatial_compression_ratio=v.spatial_compression_ratio, temporal_compression_ratio=v.temporal_compression_ratio)
    return LTXVFullConfig(
        version=p.version.decode(), guidance_scale=p.guidance_scale, num_inference_steps=p.num_inference_steps, stg_scale=p.stg_scale,
        rescaling_scale=p.rescaling_scale, stochastic_sampling=bool(p.stochastic_sampling), skip_block_list=list(p.skip_block_list[:p.n_skip_blocks]),
        timesteps=[float(x) for x in p.timesteps[:p.n_timesteps]] if p.n_timesteps else None,
        decode_timestep=float(p.decode_timestep) if p.has_decode_timestep else None,
        decode_noise_scale=float(p.decode_noise_scale) if p.has_decode_noise_scale else None,
        transformer=tcfg, vae=vcfg, vae_encoder_block_out_channels=list(p.vae_encoder_block_out_channels),
        vae_encoder_layers_per_block=list(p.vae_encoder_layers_per_block),
        scheduler=dict(num_train_timesteps=p.num_train_timesteps, shift=p.shift, use_dynamic_shifting=bool(p.use_dynamic_shifting),
                       base_shift=p.base_shift, max_shift=p.max_shift, base_image_seq_len=p.base_image_seq_len, max_image_seq_len=p.max_image_seq_len,
                       shift_terminal=p.shift_terminal if p.has_shift_terminal else None, time_shift_type="exponential" if p.time_shift_exponential else "linear"))


# ------------------------------------------------------------------ start-up control (include/ltxhip.h)
def warmup(transformer: Optional[LtxVideoTransformer3DModel], vae: Optional["AutoencoderKLLtxVideo"], B: int, F: int, H: int, W: int, K: int = 128):
    """ltx_warmup: plan measurement / workspace sizing for the latent geometry (F, H, W), outside the first real call."""
    _check(lib.ltx_warmup(transformer._h if transformer is not None else None, vae._h if vae is not None else None, B, F, H, W, K, _stream()))


def set_autotune(enabled: bool):
    _check(lib.ltx_set_autotune(int(enabled)))


def set_option(key: str, value=None):
    """ltx_set_option (include/ltxhip.h, run-time options): value None = the option's default"""
    _check(lib.ltx_set_option(key.encode(), None if value is None else str(value).encode()))


def reset_options():
    _check(lib.ltx_reset_options())


def has_experiments() -> bool:
    return bool(lib.ltx_has_experiments())


def attention_fallback_counts(reset: bool = False):
    """(head_dim-64 workgroups, head_dim-128 launches) that took the exact-max second pass since the last reset"""
    out = (C.c_ulonglong * 2)()
    _check(lib.ltx_attention_fallback_counts(out, int(reset)))
    return int(out[0]), int(out[1])


def get_option(key: str):
    """ltx_get_option: the option's current value as text (None: an experiment knob that is not set)"""
    buf = C.create_string_buffer(256)
    if lib.ltx_get_option(key.encode(), buf, 256) != 0:
        if key.startswith("x_"):
            return None
        _check(1)
    return buf.value.decode()


class options:
    """with ltxhip.options(gemm_tune=0, gemm_off="asm16"): ...  - options set for the block, the PREVIOUS values (LTX_OPTIONS'
    or an outer block's) restored after it"""
    def __init__(self, **kw): self.kw = kw
    def __enter__(self):
        self.prev = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items(): set_option(k, v)
        return self
    def __exit__(self, *a):
        for k, v in self.prev.items(): set_option(k, v)
        return False


def plan_save(path: str):
    _check(lib.ltx_plan_save(path.encode()))


def plan_load(path: str):
    _check(lib.ltx_plan_load(path.encode()))


# ------------------------------------------------------------------ frame output (include/ltxhip_frames.h)
def video_to_rgb8(video: torch.Tensor) -> torch.Tensor:
    """[B,3,F,H,W] f32 (0..255) -> [B,F,H,W,3] u8 on the device (main.rs:659-664: permute, clamp, truncating cast)."""
    v = _dev(video, torch.float32)
    B, _, F, H, W = v.shape
    out = torch.empty(B, F, H, W, 3, dtype=torch.uint8, device=v.device)
    _check(lib.ltx_video_to_rgb8(_ptr(v), B, F, H, W, _ptr(out), _stream()))
    return out


def write_png(path: str, rgb: torch.Tensor):
    """HOST u8 [H,W,3] -> PNG file."""
    t = rgb.detach().to("cpu", torch.uint8).contiguous()
    _check(lib.ltx_write_png(path.encode(), C.c_void_p(t.data_ptr()), t.shape[1], t.shape[0]))


def write_gif(path: str, frames: torch.Tensor, delay_cs: int = 4, speed: int = 30):
    """HOST u8 [N,H,W,3] -> animated GIF (main.rs:683-707: per-frame NeuQuant palette at `speed`, delay, infinite loop)."""
    t = frames.detach().to("cpu", torch.uint8).contiguous()
    _check(lib.ltx_write_gif(path.encode(), C.c_void_p(t.data_ptr()), t.shape[0], t.shape[2], t.shape[1], delay_cs, speed))


def save_video_gif(video: torch.Tensor, path: str):
    """The reference's default output: [B,3,F,H,W] f32 (0..255) on the device -> path (video.gif in main.rs:690)."""
    v = _dev(video, torch.float32)
    B, _, F, H, W = v.shape
    _check(lib.ltx_save_video_gif(_ptr(v), B, F, H, W, path.encode(), _stream()))


def save_frames_png(video: torch.Tensor, out_dir: str) -> int:
    """`--frames` output of examples/ltx-video/main.rs:653-675: out_dir/frame_%04d.png; returns the number written."""
    v = _dev(video, torch.float32)
    B, _, F, H, W = v.shape
    n = C.c_int(0)
    _check(lib.ltx_save_frames_png(_ptr(v), B, F, H, W, out_dir.encode(), C.byref(n), _stream()))
    return n.value


# ------------------------------------------------------------------ kernel-level ops (include/ltxhip_ops.h)
class ops:
    @staticmethod
    def linear(x, w, bias=None, epi=0, resid=None, gate=None, rows_per_batch=1):
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        _check(lib.ltx_op_linear(_ptr(x.contiguous()), _ptr(w.contiguous()), _ptr(bias), _ptr(y), M, N, K, _dt(x.dtype), epi,
                                 _ptr(resid), _ptr(gate), rows_per_batch, _stream()))
        return y

    @staticmethod
    def ring_pack(w):
        """-> the tile-contiguous second copy of a bf16 [N, K] weight that the small-M kernel streams (GemmArgs::Wp)"""
        N, K = w.shape
        out = torch.empty(lib.ltx_op_ring_packed_bytes(N, K) // 2, dtype=torch.bfloat16, device=w.device)
        _check(lib.ltx_op_ring_pack(_ptr(w.contiguous()), N, K, _ptr(out), _stream()))
        return out

    @staticmethod
    def linear_packed(x, w, wp, bias=None, epi=0, resid=None, gate=None, rows_per_batch=1):
        """ops.linear with the packed copy of w at hand (bf16; same results)"""
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        _check(lib.ltx_op_linear_packed(_ptr(x.contiguous()), _ptr(w.contiguous()), _ptr(wp), _ptr(bias), _ptr(y), M, N, K, epi,
                                        _ptr(resid), _ptr(gate), rows_per_batch, _stream()))
        return y

    @staticmethod
    def linear_rowsq(x, w, bias, epi=0, resid=None, gate=None, rows_per_batch=1):
        """-> (x @ w^T + bias, per-row partial sums of squares of that output per 128-column group [M, ceil(N/128)] f32)"""
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        rs = torch.empty(M, (N + 127) // 128, dtype=torch.float32, device=x.device)
        _check(lib.ltx_op_linear_rowsq(_ptr(x.contiguous()), _ptr(w.contiguous()), _ptr(bias), _ptr(y), _ptr(rs), M, N, K, _dt(x.dtype), epi,
                                       _ptr(resid), _ptr(gate), rows_per_batch, _stream()))
        return y, rs

    @staticmethod
    def rowsq(x):
        """the stand-alone form of linear_rowsq's by-product on a stored matrix (same values, bit for bit)"""
        M, N = x.shape
        rs = torch.empty(M, (N + 127) // 128, dtype=torch.float32, device=x.device)
        _check(lib.ltx_op_rowsq(_ptr(x.contiguous()), C.c_int64(M), N, N, _ptr(rs), _dt(x.dtype), _stream()))
        return rs

    @staticmethod
    def attention_rowsq(q, k, v, heads, scale, key_bias, q_rowsq, eps=1e-5):
        """cross attention on UN-normalised queries: the RMS-norm scalar of each query row comes from q_rowsq [B*Sq, D/128]"""
        B, Sq, D = q.shape
        o = torch.empty_like(q)
        _check(lib.ltx_op_attention_rowsq(_ptr(q.contiguous()), _ptr(k.contiguous()), _ptr(v.contiguous()), _ptr(o), B, Sq, k.shape[1], heads, D // heads,
                                          D, D, D, D, C.c_float(scale), _ptr(key_bias), _ptr(q_rowsq), q_rowsq.shape[-1], D, C.c_float(eps), _stream()))
        return o

    @staticmethod
    def attention_compact(q, k, v, heads, scale, key_bias, q_rowsq=None, eps=1e-5):
        """ops.attention (bf16, head_dim 64, <= 128 keys, key bias) the way the DiT's cross attention runs it: keys whose bias is
        above -5000 compacted to the front, only their key blocks multiplied.  Returns (o, keys kept per batch row)."""
        B, Sq, D = q.shape
        o = torch.empty_like(q)
        cnt = torch.zeros(B, dtype=torch.int32, device=q.device)
        _check(lib.ltx_op_attention_compact(_ptr(q.contiguous()), _ptr(k.contiguous()), _ptr(v.contiguous()), _ptr(o), B, Sq, k.shape[1], heads, D // heads,
                                            D, D, D, D, C.c_float(scale), _ptr(key_bias.contiguous()), _ptr(q_rowsq) if q_rowsq is not None else None,
                                            q_rowsq.shape[-1] if q_rowsq is not None else 0, D, C.c_float(eps), _ptr(cnt), _stream()))
        return o, cnt

    @staticmethod
    def linear_segmented(x, w, bias, seg_width):
        """x @ w^T + bias written as N/seg_width dense [M, seg_width] matrices (the DiT's q|k|v projection)."""
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(N // seg_width, M, seg_width, dtype=x.dtype, device=x.device)
        _check(lib.ltx_op_linear_segmented(_ptr(x.contiguous()), _ptr(w.contiguous()), _ptr(bias), _ptr(y), M, N, K, seg_width,
                                           _dt(x.dtype), _stream()))
        return y

    @staticmethod
    def rownorm(x, kind=0, eps=1e-6, weight=None, scale=None, shift=None, rows_per_batch=1, act=0):
        rows, D = x.shape
        y = torch.empty_like(x)
        ms = scale.shape[-1] if scale is not None else 0
        _check(lib.ltx_op_rownorm(_ptr(x.contiguous()), _ptr(y), C.c_int64(rows), D, kind, C.c_float(eps), _ptr(weight),
                                  _ptr(scale), _ptr(shift), C.c_int64(rows_per_batch), ms, act, _dt(x.dtype), _stream()))
        return y

    @staticmethod
    def linear_deferred(x, w):
        """x @ w^T as UN-reduced K ranges (the library's shape rule): parts [P, M, N] f32 (bf16 layers of at most 512 rows)"""
        M, K = x.shape
        N = w.shape[0]
        P = lib.ltx_op_linear_split_factor(M, N, K)
        parts = torch.empty(P, M, N, dtype=torch.float32, device=x.device)
        _check(lib.ltx_op_linear_deferred(_ptr(x.contiguous()), _ptr(w.contiguous()), _ptr(parts), M, N, K, _stream()))
        return parts

    @staticmethod
    def rownorm_deferred(parts, bias, resid, gate=None, kind=0, eps=1e-6, scale=None, shift=None, rows_per_batch=1):
        """finish the rows of ops.linear_deferred (h = resid + gate * (sum of the parts + bias)) and normalise them: (h, y)"""
        P, rows, D = parts.shape
        h = torch.empty_like(resid); y = torch.empty_like(resid)
        ms = scale.shape[-1] if scale is not None else 0
        _check(lib.ltx_op_rownorm_deferred(_ptr(parts), P, _ptr(bias), _ptr(resid.contiguous()), _ptr(gate), gate.shape[-1] if gate is not None else 0, _ptr(h),
                                           _ptr(y), C.c_int64(rows), D, kind, C.c_float(eps), _ptr(scale), _ptr(shift), C.c_int64(rows_per_batch), ms, _dt(resid.dtype), _stream()))
        return h, y

    @staticmethod
    def rownorm_presum(x, presum, eps=1e-6, weight=None, scale=None, shift=None, rows_per_batch=1, act=0):
        """RMS rownorm of rows whose sums of squares are known (presum [rows, D/128] from linear_rowsq / rowsq): a pure map"""
        rows, D = x.shape
        y = torch.empty_like(x)
        ms = scale.shape[-1] if scale is not None else 0
        _check(lib.ltx_op_rownorm_presum(_ptr(x.contiguous()), _ptr(y), C.c_int64(rows), D, C.c_float(eps), _ptr(weight), _ptr(scale), _ptr(shift),
                                         C.c_int64(rows_per_batch), ms, act, _ptr(presum), presum.shape[-1], _dt(x.dtype), _stream()))
        return y

    @staticmethod
    def qknorm_rope(x, weight, eps=1e-5, cos=None, sin=None):
        x = x.clone().contiguous()
        rows, D = x.shape
        _check(lib.ltx_op_qknorm_rope(_ptr(x), C.c_int64(rows), D, D, _ptr(weight), C.c_float(eps), _ptr(cos), _ptr(sin),
                                      _dt(x.dtype), _stream()))
        return x

    @staticmethod
    def timestep_embedding(timesteps, vae_flavour=False, multiplier=1.0, dtype=torch.float32, device="cuda"):
        """get_timestep_embedding(dim 256, [cos | sin]): the DiT's (ltx_transformer.rs:271-309) or the VAE's (vae.rs:172-198)"""
        ts = [float(t) for t in timesteps]
        out = torch.empty(len(ts), 256, dtype=dtype, device=device)
        _check(lib.ltx_op_timestep_embedding(_floats(ts), len(ts), int(bool(vae_flavour)), C.c_float(multiplier), _dt(dtype), _ptr(out), _stream()))
        return out

    @staticmethod
    def rope_table(B, F, H, W, D, coords=None, rope_scale=None, device="cuda"):
        cos = torch.empty(B * F * H * W, D // 2, dtype=torch.float32, device=device)
        sin = torch.empty_like(cos)
        _check(lib.ltx_op_rope_table(_ptr(cos), _ptr(sin), _ptr(coords), B, F, H, W, D,
                                     _floats(rope_scale) if rope_scale is not None else None, _stream()))
        return cos, sin

    @staticmethod
    def attention(q, k, v, heads, scale, key_bias=None):
        B, Sq, D = q.shape
        Sk = k.shape[1]
        o = torch.empty_like(q)
        _check(lib.ltx_op_attention(_ptr(q.contiguous()), _ptr(k.contiguous()), _ptr(v.contiguous()), _ptr(o), B, Sq, Sk, heads, D // heads,
                                    D, D, D, D, C.c_float(scale), _ptr(key_bias), _dt(q.dtype), _stream()))
        return o

    @staticmethod
    def attention_prescaled(q, k, v, heads):
        """bf16, head_dim 64: q already multiplied by scale*log2(e); softmax in base 2 (DiT self-attention fast path)."""
        B, Sq, D = q.shape
        def rows(t):          # a [B, S, D] view whose rows are dense and whose batches are S rows apart keeps its row stride (column slices of a fused buffer)
            return t if t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) % 8 == 0 else t.contiguous()
        q, k, v = rows(q), rows(k), rows(v)
        o = torch.empty(B, Sq, D, dtype=q.dtype, device=q.device)
        _check(lib.ltx_op_attention_prescaled(_ptr(q), _ptr(k), _ptr(v), _ptr(o), B, Sq, k.shape[1],
                                              heads, D // heads, q.stride(1), k.stride(1), v.stride(1), D, _stream()))
        return o

    @staticmethod
    def conv3d(x_cl, w, bias, causal=False, resid=None):
        B, T, H, W, Cin = x_cl.shape
        Cout = w.shape[0]
        y = torch.empty(B, T, H, W, Cout, dtype=x_cl.dtype, device=x_cl.device)
        _check(lib.ltx_op_conv3d(_ptr(x_cl.contiguous()), _ptr(w.contiguous()), _ptr(bias.to(w.dtype).contiguous()), _dt(w.dtype), _ptr(y), _ptr(resid),
                                 B, T, H, W, Cin, Cout, int(causal), _dt(x_cl.dtype), _stream()))
        return y

    @staticmethod
    def upsample3d(x_cl, w, bias, causal=False, residual=True):
        B, T, H, W, Cin = x_cl.shape
        Cout = w.shape[0]
        y = torch.empty(B, 2 * T - 1, 2 * H, 2 * W, Cout // 8, dtype=x_cl.dtype, device=x_cl.device)
        _check(lib.ltx_op_upsample3d(_ptr(x_cl.contiguous()), _ptr(w.contiguous()), _ptr(bias.to(w.dtype).contiguous()), _dt(w.dtype), _ptr(y),
                                     B, T, H, W, Cin, Cout, int(causal), int(residual), _dt(x_cl.dtype), _stream()))
        return y

    @staticmethod
    def gemm_plan(M, N, K, conv=0, ntaps=1, T=1, H=1, W=1) -> str:
        """name of the GEMM plan the dispatcher cached for this bf16 shape ("" before its first run)."""
        buf = C.create_string_buffer(32)
        _check(lib.ltx_op_gemm_plan(M, N, K, conv, ntaps, T, H, W, buf, 32))
        return buf.value.decode()

    @staticmethod
    def blend(a, b, dim, blend_extent):
        """in place on a copy of b; a, b f32 [B,C,t,h,w]."""
        a = a.contiguous(); b = b.clone().contiguous()
        BC = a.shape[0] * a.shape[1]
        _check(lib.ltx_op_blend(_ptr(a), _ptr(b), BC, a.shape[2], a.shape[3], a.shape[4], b.shape[2], b.shape[3], b.shape[4],
                                dim, blend_extent, _stream()))
        return b

    @staticmethod
    def conv_out_unpatchify(x_cl, w, bias, causal=False, postprocess=False):
        B, T, H, W, Cin = x_cl.shape
        Cout = w.shape[0]
        y = torch.empty(B, Cout // 16, T, 4 * H, 4 * W, dtype=torch.float32, device=x_cl.device)
        _check(lib.ltx_op_conv_out_unpatchify(_ptr(x_cl.contiguous()), _ptr(w.contiguous()), _ptr(bias.to(w.dtype).contiguous()), _dt(w.dtype), _ptr(y),
                                              B, T, H, W, Cin, Cout, int(causal), int(postprocess), _dt(x_cl.dtype), _stream()))
        return y


def prof_enable(on: bool):
    _check(lib.ltx_prof_enable(int(on)))


def prof_report(kind: int):
    """-> (total_ms, total_work, count) for kernel class `kind` since prof_enable(True)."""
    ms, work, cnt = C.c_double(), C.c_double(), C.c_longlong()
    _check(lib.ltx_prof_report(kind, C.byref(ms), C.byref(work), C.byref(cnt)))
    return ms.value, work.value, cnt.value


PROF_KERNELS = ("gemm_kernel (128 x 128)", "gemm_big_kernel", "gemm_p8_kernel", "conv_halo_kernel", "gemm_asm_kernel (32x32x16)", "gemm_asm16_kernel", "gemm_ring_kernel")


def prof_report_kernel(kind: int, kernel: int):
    """-> (total_ms, total_work, count) of the launches of class `kind` served by kernel `kernel` (index into PROF_KERNELS)."""
    ms, work, cnt = C.c_double(), C.c_double(), C.c_longlong()
    _check(lib.ltx_prof_report_kernel(kind, kernel, C.byref(ms), C.byref(work), C.byref(cnt)))
    return ms.value, work.value, cnt.value
