"""CPU oracle for the LTX-Video hot path (DiT denoise step + 3D-VAE decode).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
shipped path (`candle-video_amd/`) never does and fails loudly when its HIP
library is missing.

It is an op-for-op restatement, in torch-CPU, of the *Rust* reference's
semantics (FerrisMind/candle-video; all `file:line` citations are relative to
the reference checkout), including the reference's quirks (integer-truncated
timesteps, bf16-rounded timestep in bf16 mode, -10000 mask bias, LayerNorm in
model dtype, RMSNorm eps values, un-fused per-op rounding).  Every function is
dtype-generic: `dtype=torch.float32` reproduces the reference's CPU parity
path; `dtype=torch.bfloat16` reproduces the per-op bf16 rounding of the
reference's GPU path (candle rounds to bf16 after every op exactly as torch's
CPU bf16 kernels do: compute in f32, round once).

PARITY PINNING STATUS
---------------------
The reference is Rust (candle) and cannot be compiled here (no cargo/rustc;
candle-core/candle-nn 0.9.2 are un-vendored), every `verify_*_parity` fixture
(`gen_*.safetensors`) is absent from the checkout, and its Python capture
scripts need `diffusers` + real checkpoints (absent).  Therefore:

  * guidance mix / latent (de)normalisation are PINNED against outputs of the
    reference's own runnable torch-only scripts (`scripts/gen_guidance_ref.py`,
    `scripts/gen_latent_norm_ref.py`), regenerated in-container by
    `tools/gen_fixtures.py` and committed under `tests/golden/`;
  * closed-form reference tests are PINNED as known-answer tests
    (`tests/verify_rope_parity.rs:646-733` AdaLN 0.102, `:473-511` attention
    scale 0.125, `configs.rs:289-324`, `t2v_pipeline.rs:159-169` mu values,
    upsampler axis-order KAT modelled on `tests/vae_tests.rs:119-180`);
  * every composite primitive is cross-checked against an INDEPENDENT torch
    library implementation (F.conv3d with replicate/zero padding, F.scaled_dot_
    product_attention, F.rms_norm, F.layer_norm, F.gelu(tanh), F.pixel_shuffle
    style einops rearranges) in `tests/test_oracle.py`;
  * the T5 encoder restatement (the reference wraps candle-transformers' port of
    Hugging Face T5) is PINNED against `transformers.T5EncoderModel` on shared
    random weights (`tests/test_t5_cpu.py`);
  * for the full DiT forward / VAE decode numbers: **parity unpinned** against
    the reference binary (no reference-produced vectors exist anywhere).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# PCG32 host RNG  (src/utils/deterministic_rng.rs:11-81)
# --------------------------------------------------------------------------

_PCG_MULT = np.uint64(6364136223846793005)
_M64 = (1 << 64) - 1


class Pcg32:
    """PCG32 XSH-RR + Box-Muller, bit-for-bit the reference's integer stream
    (deterministic_rng.rs:11-35); floats follow its f32 arithmetic (:37-59)."""

    def __init__(self, seed: int, inc: int):
        self.state = 0
        self.inc = ((inc << 1) | 1) & _M64          # :15
        self.next_u32()                               # :17
        self.state = (self.state + seed) & _M64       # :18
        self.next_u32()                               # :19

    def next_u32(self) -> int:                        # :23-35
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & _M64
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = (old >> 59) & 31
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def _u32_block(self, n: int) -> np.ndarray:
        """n consecutive next_u32() outputs, vectorised (LCG jump via wrapped
        uint64 cumprod / cumsum); advances the state exactly as n calls would."""
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        with np.errstate(over="ignore"):
            a = np.full(n, _PCG_MULT, dtype=np.uint64)
            a[0] = np.uint64(1)
            apow = np.cumprod(a, dtype=np.uint64)                 # a^i, i=0..n-1
            geo = np.cumsum(apow, dtype=np.uint64)                 # sum_{j<=i} a^j
            s0 = np.uint64(self.state)
            inc = np.uint64(self.inc)
            # state before output i:  s_i = a^i s0 + inc * sum_{j<i} a^j
            geo_prev = np.concatenate([np.zeros(1, dtype=np.uint64), geo[:-1]])
            old = apow * s0 + inc * geo_prev
            # state after n steps
            last = old[-1] * _PCG_MULT + inc
        self.state = int(last)
        xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        with np.errstate(over="ignore"):
            out = (xorshifted >> rot) | (xorshifted << ((np.uint32(0) - rot) & np.uint32(31)))
        return out.astype(np.uint32)

    def next_f32(self) -> np.float32:                 # :37-41
        return np.float32(self.next_u32() >> 8) * np.float32(5.9604645e-8)

    def next_gaussian(self) -> Tuple[np.float32, np.float32]:   # :45-59
        while True:
            u1 = self.next_f32()
            if u1 > np.float32(1e-7):
                break
        u2 = self.next_f32()
        mag = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
        ang = np.float32(2.0) * np.float32(math.pi) * u2
        return mag * np.cos(ang, dtype=np.float32), mag * np.sin(ang, dtype=np.float32)

    def randn(self, shape: Sequence[int]) -> Tensor:  # :62-81
        n = int(np.prod(shape))
        npairs = (n + 1) // 2
        # Fast path: draw 2*npairs u32; if a rejection (u1 <= 1e-7, i.e. the
        # 24-bit mantissa is 0 or 1) appears, fall back to the scalar loop from
        # that point (probability 1.2e-7 per draw).
        save_state = self.state
        u = self._u32_block(2 * npairs)
        m = (u >> np.uint32(8))
        u1m = m[0::2]
        if np.any(u1m < 2):
            self.state = save_state
            out = np.empty(2 * npairs, dtype=np.float32)
            for i in range(npairs):
                z0, z1 = self.next_gaussian()
                out[2 * i] = z0
                out[2 * i + 1] = z1
        else:
            f = m.astype(np.float32) * np.float32(5.9604645e-8)
            u1 = f[0::2]
            u2 = f[1::2]
            mag = np.sqrt(np.float32(-2.0) * np.log(u1), dtype=np.float32)
            ang = (np.float32(2.0) * np.float32(math.pi)) * u2
            out = np.empty(2 * npairs, dtype=np.float32)
            out[0::2] = mag * np.cos(ang)
            out[1::2] = mag * np.sin(ang)
        return torch.from_numpy(out[:n].copy()).reshape(*shape)


# --------------------------------------------------------------------------
# DiT  (src/models/ltx_video/ltx_transformer.rs)
# --------------------------------------------------------------------------

@dataclass
class DitConfig:                      # ltx_transformer.rs:23-58
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

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """candle_nn::Linear: x @ w.T + b, result in x.dtype."""
    return F.linear(x, w, b)


def layer_norm_no_params(x: Tensor, eps: float = 1e-6) -> Tensor:
    """LayerNormNoParams::forward (ltx_transformer.rs:72-79): biased variance,
    computed in the tensor's own dtype (NOT up-cast)."""
    d = x.shape[-1]
    mean = x.sum(-1, keepdim=True) / d
    xc = x - mean
    var = (xc * xc).sum(-1, keepdim=True) / d
    denom = (var + eps).sqrt()
    return xc / denom


def rms_norm(x: Tensor, weight: Optional[Tensor], eps: float) -> Tensor:
    """RmsNorm::forward (ltx_transformer.rs:99-119): f32 statistics, cast back,
    then weight multiply in model dtype."""
    dt = x.dtype
    xf = x.float()
    ms = (xf * xf).sum(-1, keepdim=True) * (1.0 / xf.shape[-1])
    denom = (ms + eps).sqrt()
    y = (xf / denom).to(dt)
    if weight is not None:
        y = y * weight
    return y


def gelu_approximate(x: Tensor) -> Tensor:
    """gelu_approximate (ltx_transformer.rs:214-226), f32 internally."""
    xf = x.float()
    cube = xf * xf * xf
    inner = xf + cube * 0.044715
    scale = np.float32(math.sqrt(2.0 / math.pi))
    t = torch.tanh(inner * float(scale))
    return ((xf * (t + 1.0)) * 0.5).to(x.dtype)


def get_timestep_embedding(t: Tensor, dim: int = 256, flip_sin_to_cos: bool = True) -> Tensor:
    """ltx_transformer.rs:271-309: inv_freq_i = 1/10000^(i/half) in f32."""
    orig = t.dtype
    half = dim // 2
    i = np.arange(half, dtype=np.float32)
    inv = (np.float32(1.0) / np.power(np.float32(10000.0), i / np.float32(half), dtype=np.float32)).astype(np.float32)
    freqs = t.float().unsqueeze(1) * torch.from_numpy(inv).unsqueeze(0)
    s, c = freqs.sin(), freqs.cos()
    emb = torch.cat([c, s], -1) if flip_sin_to_cos else torch.cat([s, c], -1)
    return emb.to(orig)


def apply_rotary_emb(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """apply_rotary_emb (ltx_transformer.rs:314-339): interleaved pairs, f32."""
    dt = x.dtype
    xf = x.float()
    b, s, c = xf.shape
    x2 = xf.reshape(b, s, c // 2, 2)
    xr, xi = x2[..., 0], x2[..., 1]
    rot = torch.stack([-xi, xr], -1).reshape(b, s, c)
    return (xf * cos.float() + rot * sin.float()).to(dt)


def rope_cos_sin(dim: int, batch: int, num_frames: int, height: int, width: int,
                 rope_interpolation_scale: Optional[Tuple[float, float, float]] = None,
                 video_coords: Optional[Tensor] = None,
                 base=(20, 2048, 2048), patch_size: int = 1, patch_size_t: int = 1,
                 theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """LtxVideoRotaryPosEmbed::forward (ltx_transformer.rs:436-524) incl.
    prepare_video_coords (:373-433).  Returns (cos, sin) f32 [B, S, dim]."""
    if video_coords is not None:
        vc = video_coords.float()
        cf = vc[..., 0] * float(np.float32(1.0) / np.float32(base[0]))   # affine(1/base_f)
        ch = vc[..., 1] * float(np.float32(1.0) / np.float32(base[1]))
        cw = vc[..., 2] * float(np.float32(1.0) / np.float32(base[2]))
        grid = torch.stack([cf, ch, cw], -1)
    else:
        gf = torch.arange(num_frames, dtype=torch.float32).reshape(-1, 1, 1).expand(num_frames, height, width)
        gh = torch.arange(height, dtype=torch.float32).reshape(1, -1, 1).expand(num_frames, height, width)
        gw = torch.arange(width, dtype=torch.float32).reshape(1, 1, -1).expand(num_frames, height, width)
        grid = torch.stack([gf, gh, gw], 0).unsqueeze(0).expand(batch, 3, num_frames, height, width)
        if rope_interpolation_scale is not None:
            sf, sh, sw = rope_interpolation_scale
            fs = float(np.float32(sf * patch_size_t / base[0]))
            hs = float(np.float32(sh * patch_size / base[1]))
            ws = float(np.float32(sw * patch_size / base[2]))
            grid = torch.cat([grid[:, 0:1] * fs, grid[:, 1:2] * hs, grid[:, 2:3] * ws], 1)
        grid = grid.reshape(batch, 3, -1).transpose(1, 2).contiguous()
    steps = dim // 6
    if steps <= 1:
        lin = torch.zeros(1, dtype=torch.float32)
    else:
        lin = torch.arange(steps, dtype=torch.float32) * (1.0 / (steps - 1))
    theta_ln = float(np.float32(math.log(theta)))
    freqs = (lin * theta_ln).exp() * (math.pi / 2.0)
    g = grid.float().unsqueeze(-1) * 2.0 - 1.0                      # [B,S,3,1]
    fr = g * freqs.reshape(1, 1, 1, steps)                          # [B,S,3,steps]
    fr = fr.transpose(-1, -2).contiguous().flatten(2)               # [B,S,steps*3]
    cos = fr.cos().repeat_interleave(2, -1)
    sin = fr.sin().repeat_interleave(2, -1)
    rem = dim % 6
    if rem:
        b, s, _ = cos.shape
        cos = torch.cat([torch.ones(b, s, rem), cos], -1)
        sin = torch.cat([torch.zeros(b, s, rem), sin], -1)
    return cos, sin


ATTN_SCORE_BYTES = 8 << 30      # host budget for one pass of f32 attention scores (attention(): heads per pass)


def attention(p: Dict[str, Tensor], prefix: str, heads: int, hidden: Tensor, enc: Optional[Tensor],
              mask_bias: Optional[Tensor], rope: Optional[Tuple[Tensor, Tensor]]) -> Tensor:
    """LtxAttention::forward manual (CPU / masked) path, ltx_transformer.rs:648-750."""
    b, q_len, _ = hidden.shape
    e = hidden if enc is None else enc
    k_len = e.shape[1]
    q = linear(hidden, p[prefix + "to_q.weight"], p.get(prefix + "to_q.bias"))
    k = linear(e, p[prefix + "to_k.weight"], p.get(prefix + "to_k.bias"))
    v = linear(e, p[prefix + "to_v.weight"], p.get(prefix + "to_v.bias"))
    q = rms_norm(q, p[prefix + "norm_q.weight"], 1e-5)
    k = rms_norm(k, p[prefix + "norm_k.weight"], 1e-5)
    if rope is not None:
        q = apply_rotary_emb(q, *rope)
        k = apply_rotary_emb(k, *rope)
    hd = q.shape[-1] // heads
    dt = q.dtype
    qf = q.reshape(b, q_len, heads, hd).transpose(1, 2).contiguous().float()
    kf = k.reshape(b, k_len, heads, hd).transpose(1, 2).contiguous().float()
    vf = v.reshape(b, k_len, heads, hd).transpose(1, 2).contiguous().float()
    scale = float(np.float32(1.0) / np.sqrt(np.float32(hd)))
    # heads per pass: all of them unless the f32 score tensor would not fit the host (C5's S = 17556: 1.2 GB per head, 39 GB
    # for 32); heads never interact and every head sees the same op sequence either way
    hc = max(1, min(heads, int(ATTN_SCORE_BYTES // max(1, b * q_len * k_len * 4))))
    outs = []
    for h0 in range(0, heads, hc):
        att = qf[:, h0:h0 + hc] @ kf[:, h0:h0 + hc].transpose(-1, -2)
        att = att * scale
        if mask_bias is not None:                   # [B,1,K] -> [B,1,1,K]  (:627-641)
            att = att + mask_bias.float().unsqueeze(2)
        att = torch.softmax(att, -1)
        outs.append(att @ vf[:, h0:h0 + hc])
        del att
    out = (outs[0] if len(outs) == 1 else torch.cat(outs, 1)).to(dt)
    out = out.transpose(1, 2).contiguous().reshape(b, q_len, heads * hd)
    return linear(out, p[prefix + "to_out.0.weight"], p.get(prefix + "to_out.0.bias"))


def transformer_block(p: Dict[str, Tensor], prefix: str, cfg: DitConfig, h: Tensor, enc: Tensor,
                      temb: Tensor, rope, mask_bias) -> Tensor:
    """LtxVideoTransformerBlock::forward (ltx_transformer.rs:820-937)."""
    b = h.shape[0]
    dim = temb.shape[-1] // 6
    n = rms_norm(h, None, cfg.norm_eps)
    ada = p[prefix + "scale_shift_table"].unsqueeze(0).unsqueeze(0) + temb.reshape(b, 1, 6, dim)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = [ada[:, :, i] for i in range(6)]
    n = n * (1 + scale_msa) + shift_msa
    a1 = attention(p, prefix + "attn1.", cfg.num_attention_heads, n, None, None, rope)
    h = h + a1 * gate_msa
    a2 = attention(p, prefix + "attn2.", cfg.num_attention_heads, h, enc, mask_bias, None)
    h = h + a2
    m = rms_norm(h, None, cfg.norm_eps)
    m = m * (1 + scale_mlp) + shift_mlp
    f = gelu_approximate(linear(m, p[prefix + "ff.net.0.proj.weight"], p[prefix + "ff.net.0.proj.bias"]))
    f = linear(f, p[prefix + "ff.net.2.weight"], p[prefix + "ff.net.2.bias"])
    return h + f * gate_mlp


def dit_forward(p: Dict[str, Tensor], cfg: DitConfig, hidden: Tensor, enc: Tensor, timestep: Tensor,
                enc_mask: Optional[Tensor], num_frames: int, height: int, width: int,
                rope_interpolation_scale=None, video_coords: Optional[Tensor] = None,
                skip_layer_mask: Optional[Tensor] = None, skip_block_list: Sequence[int] = (),
                dtype: torch.dtype = torch.float32) -> Tensor:
    """LtxVideoTransformer3DModel::forward (ltx_transformer.rs:1029-1172).
    `p` holds weights already in `dtype`."""
    dt = dtype
    h = hidden.to(dt)
    enc = enc.to(dt)
    h = linear(h, p["proj_in.weight"], p["proj_in.bias"])
    t = timestep.flatten().to(dt)                                  # :1051 (bf16-rounds the timestep)
    # AdaLayerNormSingle (:262-267)
    tproj = get_timestep_embedding(t, 256, True)
    e = linear(tproj, p["time_embed.emb.timestep_embedder.linear_1.weight"], p["time_embed.emb.timestep_embedder.linear_1.bias"])
    e = F.silu(e)
    embedded_timestep = linear(e, p["time_embed.emb.timestep_embedder.linear_2.weight"], p["time_embed.emb.timestep_embedder.linear_2.bias"])
    temb = linear(F.silu(embedded_timestep), p["time_embed.linear.weight"], p["time_embed.linear.bias"])
    # caption projection (:186-190)
    c = linear(enc, p["caption_projection.linear_1.weight"], p["caption_projection.linear_1.bias"])
    c = gelu_approximate(c)
    enc = linear(c, p["caption_projection.linear_2.weight"], p["caption_projection.linear_2.bias"])
    mask_bias = None
    if enc_mask is not None:                                       # :1059-1070
        mf = enc_mask.to(h.dtype)
        mask_bias = ((mf * -1.0 + 1.0) * (-10000.0)).unsqueeze(1)
    cos, sin = rope_cos_sin(cfg.inner_dim, h.shape[0], num_frames, height, width,
                            rope_interpolation_scale, video_coords,
                            patch_size=cfg.patch_size, patch_size_t=cfg.patch_size_t)
    for idx in range(cfg.num_layers):
        if idx in skip_block_list:
            continue
        orig = h
        h = transformer_block(p, f"transformer_blocks.{idx}.", cfg, h, enc, temb, (cos, sin), mask_bias)
        if skip_layer_mask is not None:                            # :1112-1123
            m = skip_layer_mask[idx].flatten().reshape(h.shape[0], 1, 1).to(h.dtype)
            h = h * (m * -1.0 + 1.0) + orig * m
    table = p["scale_shift_table"].to(embedded_timestep.dtype).unsqueeze(0).unsqueeze(0)
    ss = table + embedded_timestep.unsqueeze(1).unsqueeze(2)       # [B,1,2,D]
    shift, scale = ss[:, :, 0], ss[:, :, 1]
    h = layer_norm_no_params(h, 1e-6)
    h = h * (1 + scale) + shift
    return linear(h, p["proj_out.weight"], p["proj_out.bias"])


def dit_weight_shapes(cfg: DitConfig) -> Dict[str, Tuple[int, ...]]:
    """Weight names/shapes of LtxVideoTransformer3DModel::new (ltx_transformer.rs:957-1003)."""
    D = cfg.inner_dim
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, i, o):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)
    lin("proj_in", cfg.in_channels, D)
    s["scale_shift_table"] = (2, D)
    lin("time_embed.emb.timestep_embedder.linear_1", 256, D)
    lin("time_embed.emb.timestep_embedder.linear_2", D, D)
    lin("time_embed.linear", D, 6 * D)
    lin("caption_projection.linear_1", cfg.caption_channels, D)
    lin("caption_projection.linear_2", D, D)
    for i in range(cfg.num_layers):
        pre = f"transformer_blocks.{i}."
        for a, kvdim in (("attn1", D), ("attn2", cfg.cross_attention_dim)):
            lin(pre + a + ".to_q", D, D)
            lin(pre + a + ".to_k", kvdim, D)
            lin(pre + a + ".to_v", kvdim, D)
            lin(pre + a + ".to_out.0", D, D)
            s[pre + a + ".norm_q.weight"] = (D,)
            s[pre + a + ".norm_k.weight"] = (D,)
        lin(pre + "ff.net.0.proj", D, 4 * D)
        lin(pre + "ff.net.2", 4 * D, D)
        s[pre + "scale_shift_table"] = (6, D)
    lin("proj_out", D, cfg.out_channels)
    return s


# --------------------------------------------------------------------------
# VAE decoder  (src/models/ltx_video/vae.rs)
# --------------------------------------------------------------------------

@dataclass
class VaeConfig:                      # vae.rs:32-103 (decoder-side fields)
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
    decoder_upsample_residual: Tuple[bool, ...] = (True, True, True)     # vae.rs:52-53, 88; reversed like the other lists (:1516-1517)
    decoder_inject_noise: Tuple[bool, ...] = (False, False, False, False)    # vae.rs:51, 87; reversed (:1514-1515): entry [-1] is the mid block
    decoder_spatiotemporal_scaling: Tuple[bool, ...] = (True, True, True)    # vae.rs:40-41, 78; reversed (:1509-1510); False = the (1, 2, 2) upsampler (:1225-1236)
    # tiling (vae.rs:1849-1854)
    tile_sample_min_height: int = 512
    tile_sample_min_width: int = 512
    tile_sample_min_num_frames: int = 16
    tile_sample_stride_height: int = 384
    tile_sample_stride_width: int = 384
    tile_sample_stride_num_frames: int = 8

    def stage_channels(self) -> List[int]:
        boc = list(reversed(self.decoder_block_out_channels))
        upf = list(reversed(self.decoder_upsample_factor))
        return [boc[i] // upf[i] for i in range(len(boc))]          # vae.rs:1548

    @property
    def mid_channels(self) -> int:
        return list(reversed(self.decoder_block_out_channels))[0]


def vae_timestep_embedding(t: Tensor, dim: int = 256) -> Tensor:
    """vae.rs:172-198: exp(-ln(1e4) * i / half), order [cos, sin], cast to t.dtype."""
    half = dim // 2
    coef = -math.log(10000.0) / half
    emb = (torch.arange(half, dtype=torch.float32) * coef).exp()
    e = t.float().unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat([e.cos(), e.sin()], 1).to(t.dtype)


def time_embedder(p, prefix: str, t: Tensor, hidden_dtype) -> Tensor:
    """CombinedTimestepEmbedder::forward (vae.rs:247-264) + TimestepEmbedder (:214-231)."""
    proj = vae_timestep_embedding(t, 256).to(hidden_dtype)
    h = linear(proj, p[prefix + "timestep_embedder.linear_1.weight"], p[prefix + "timestep_embedder.linear_1.bias"])
    h = F.silu(h)
    return linear(h, p[prefix + "timestep_embedder.linear_2.weight"], p[prefix + "timestep_embedder.linear_2.bias"])


def causal_conv3d(x: Tensor, w: Tensor, b: Tensor, is_causal: bool) -> Tensor:
    """LtxVideoCausalConv3d::forward (vae.rs:374-465): replicate-pad T, then for
    every output frame the sum over kt of conv2d slices (zero pad kh//2 on H and
    W, :337-349), bias added once after the sum."""
    kt, kh = w.shape[2], w.shape[3]
    if kt > 1:
        if is_causal:
            x = torch.cat([x[:, :, :1].repeat(1, 1, kt - 1, 1, 1), x], 2)
        else:
            l = (kt - 1) // 2
            x = torch.cat([x[:, :, :1].repeat(1, 1, l, 1, 1), x, x[:, :, -1:].repeat(1, 1, l, 1, 1)], 2)
    t_pad = x.shape[2]
    t_out = t_pad - (kt - 1)
    ys = []
    for to in range(t_out):
        acc = None
        for ki in range(kt):
            yt = F.conv2d(x[:, :, to + ki], w[:, :, ki].contiguous(), None, padding=kh // 2)
            acc = yt if acc is None else acc + yt
        ys.append(acc.unsqueeze(2))
    y = torch.cat(ys, 2)
    return y + b.reshape(1, -1, 1, 1, 1)


def rms_norm_channels_first(x: Tensor, eps: float = 1e-8) -> Tensor:
    """rmsnorm_channels_first (vae.rs:148-153) with candle_nn::RmsNorm(ones, 1e-8)
    (:618-628): f32 statistics, cast back, multiply by ones."""
    xp = x.permute(0, 2, 3, 4, 1)
    xf = xp.float()
    ms = (xf * xf).mean(-1, keepdim=True)
    y = (xf / (ms + eps).sqrt()).to(x.dtype)
    return y.permute(0, 4, 1, 2, 3)


class NoisePlanes:
    """The [H, W] planes maybe_inject_noise draws (vae.rs:741-753).  The reference takes them from the device RNG
    (Tensor::randn, unseeded: not reproducible); engine and oracle agree on plane k = Pcg32(seed, k).randn((H, W)),
    k counting the injections of a decoder's life (include/ltxhip.h ltx_vae_set_noise_seed)."""

    def __init__(self, seed: int = 0):
        self.seed, self.k = seed, 0

    def plane(self, h: int, w: int) -> Tensor:
        t = Pcg32(self.seed, self.k).randn((h, w))
        self.k += 1
        return t


def _inject_noise(h: Tensor, scale: Optional[Tensor], noise: Optional["NoisePlanes"]) -> Tensor:
    """maybe_inject_noise (vae.rs:741-753): x + noise[1,1,1,H,W].to(dtype) * scale[1,C,1,1,1]; no scale in the checkpoint = no-op."""
    if scale is None or noise is None:
        return h
    n = noise.plane(h.shape[3], h.shape[4]).to(h.dtype).reshape(1, 1, 1, h.shape[3], h.shape[4])
    return h + n * scale.reshape(1, -1, 1, 1, 1).to(h.dtype)


def resnet_block(p, prefix: str, x: Tensor, temb: Optional[Tensor], is_causal: bool,
                 noise: Optional["NoisePlanes"] = None) -> Tensor:
    """LtxVideoResnetBlock3d::forward (vae.rs:755-821), in==out channels;
    maybe_apply_scale_shift (:711-739); noise injection (:741-753, 784, 809) when `noise` is given (the block's
    inject flag) and the checkpoint holds per_channel_scaleN.weight (:676-689)."""
    tbl = p.get(prefix + "scale_shift_table")

    def mod(h, stage):
        if tbl is None or temb is None:
            return h
        b = temb.shape[0]
        c = tbl.shape[1]
        tt = temb.reshape(b, 4, c, 1, 1, 1) + tbl.unsqueeze(0).unsqueeze(3).unsqueeze(4).unsqueeze(5)
        shift, scale = tt[:, stage * 2], tt[:, stage * 2 + 1]
        return h * (scale + 1.0) + shift

    h = rms_norm_channels_first(x)
    h = mod(h, 0)
    h = F.silu(h)
    h = causal_conv3d(h, p[prefix + "conv1.conv.weight"], p[prefix + "conv1.conv.bias"], is_causal)
    h = _inject_noise(h, p.get(prefix + "per_channel_scale1.weight"), noise)
    h = rms_norm_channels_first(h)
    h = mod(h, 1)
    h = F.silu(h)
    h = causal_conv3d(h, p[prefix + "conv2.conv.weight"], p[prefix + "conv2.conv.bias"], is_causal)
    h = _inject_noise(h, p.get(prefix + "per_channel_scale2.weight"), noise)
    return h + x


def depth_to_space(x: Tensor, st: int, sh: int, sw: int) -> Tensor:
    """vae.rs:1106-1114 / 1142-1158: [B, C'*st*sh*sw, T,H,W] -> [B,C',T*st,H*sh,W*sw],
    packed channel = ((c'*st + it)*sh + ih)*sw + iw."""
    b, c, t, h, w = x.shape
    co = c // (st * sh * sw)
    x = x.reshape(b, co, st, sh, sw, t, h, w).permute(0, 1, 5, 2, 6, 3, 7, 4).contiguous()
    return x.reshape(b, co, t * st, h * sh, w * sw)


def upsampler(p, prefix: str, x: Tensor, out_channels: int, is_causal: bool,
              stride=(2, 2, 2), residual: bool = True) -> Tensor:
    """LtxVideoUpsampler3d::forward (vae.rs:1090-1169)."""
    st, sh, sw = stride
    res = None
    if residual:
        r = depth_to_space(x, st, sh, sw)
        repeats = (out_channels * st * sh * sw) // x.shape[1]       # :1068
        if repeats > 1:
            r = r.repeat(1, repeats, 1, 1, 1)
        res = r[:, :, st - 1:]
    h = causal_conv3d(x, p[prefix + "conv.conv.weight"], p[prefix + "conv.conv.bias"], is_causal)
    h = depth_to_space(h, st, sh, sw)[:, :, st - 1:]
    return h + res if res is not None else h


def unpatchify(x: Tensor, p_: int, pt: int) -> Tensor:
    """LtxVideoDecoder3d::unpatchify (vae.rs:1626-1654): permute(0,1,5,2,6,4,7,3)."""
    b, c, f, h, w = x.shape
    oc = c // (pt * p_ * p_)
    x = x.reshape(b, oc, pt, p_, p_, f, h, w).permute(0, 1, 5, 2, 6, 4, 7, 3).contiguous()
    return x.reshape(b, oc, f * pt, h * p_, w * p_)


def decoder_forward(p: Dict[str, Tensor], cfg: VaeConfig, z: Tensor, temb: Optional[Tensor],
                    dtype: torch.dtype = torch.float32, noise: Optional[NoisePlanes] = None) -> Tensor:
    """LtxVideoDecoder3d::forward (vae.rs:1656-1726); `p` keys are relative to `decoder.`."""
    causal = cfg.decoder_causal
    z = z.to(dtype)
    t = temb.to(dtype) if temb is not None else None
    h = causal_conv3d(z, p["conv_in.conv.weight"], p["conv_in.conv.bias"], causal)
    ts = None
    if t is not None:
        ts = t.flatten()
        if "timestep_scale_multiplier" in p:
            ts = ts * p["timestep_scale_multiplier"]
    # mid block (vae.rs:998-1034)
    te = None
    if ts is not None and cfg.timestep_conditioning:
        e = time_embedder(p, "mid_block.time_embedder.", ts, h.dtype)
        te = e.reshape(h.shape[0], -1, 1, 1, 1)
    nres = list(reversed(cfg.decoder_layers_per_block))
    inj = list(reversed(cfg.decoder_inject_noise))                 # vae.rs:1514-1515
    sts = list(reversed(cfg.decoder_spatiotemporal_scaling))       # vae.rs:1509-1510
    nz = lambda k: noise if (noise is not None and k < len(inj) and inj[k]) else None
    for i in range(nres[0]):
        h = resnet_block(p, f"mid_block.resnets.{i}.", h, te, causal, nz(0))
    # up blocks (vae.rs:1274-1312)
    for bi, ch in enumerate(cfg.stage_channels()):
        pre = f"up_blocks.{bi}."
        te = None
        if ts is not None and cfg.timestep_conditioning:
            e = time_embedder(p, pre + "time_embedder.", ts, h.dtype)
            te = e.reshape(h.shape[0], -1, 1, 1, 1)
        upr = list(reversed(cfg.decoder_upsample_residual))
        stride = (2, 2, 2) if (bi >= len(sts) or sts[bi]) else (1, 2, 2)      # vae.rs:1212-1236
        h = upsampler(p, pre + "upsamplers.0.", h, ch, causal, stride=stride, residual=bool(upr[bi]) if bi < len(upr) else True)
        for i in range(nres[bi + 1]):
            h = resnet_block(p, pre + f"resnets.{i}.", h, te, causal, nz(bi + 1))
    h = rms_norm_channels_first(h)
    if ts is not None and cfg.timestep_conditioning:
        e = time_embedder(p, "time_embedder.", ts, h.dtype)
        c = p["scale_shift_table"].shape[1]
        tt = e.reshape(h.shape[0], 2, c) + p["scale_shift_table"].unsqueeze(0)
        shift = tt[:, 0].reshape(h.shape[0], c, 1, 1, 1)
        scale = tt[:, 1].reshape(h.shape[0], c, 1, 1, 1)
        h = h * (scale + 1.0) + shift
    h = F.silu(h)
    h = causal_conv3d(h, p["conv_out.conv.weight"], p["conv_out.conv.bias"], causal)
    return unpatchify(h, cfg.patch_size, cfg.patch_size_t)


def _blend(a: Tensor, b: Tensor, extent: int, dim: int) -> Tensor:
    """blend_h / blend_v / blend_t (vae.rs:1927-2006)."""
    blend = min(extent, a.shape[dim], b.shape[dim])
    if blend == 0:
        return b
    w = (torch.arange(blend, dtype=torch.float32) * (1.0 / blend))
    shape = [1] * 5
    shape[dim] = blend
    w = w.reshape(shape).to(b.dtype)
    one_minus = (-w) + 1.0
    b_head = b.narrow(dim, 0, blend)
    b_tail = b.narrow(dim, blend, b.shape[dim] - blend)
    a_tail = a.narrow(dim, a.shape[dim] - blend, blend)
    mixed = a_tail * one_minus + b_head * w
    return torch.cat([mixed, b_tail], dim)


def tiled_decode(p, cfg: VaeConfig, z: Tensor, temb, dtype, noise: Optional[NoisePlanes] = None) -> Tensor:
    """AutoencoderKLLtxVideo::tiled_decode (vae.rs:2225-2290)."""
    _, _, _, height, width = z.shape
    r = cfg.spatial_compression_ratio
    tmin_h, tmin_w = cfg.tile_sample_min_height // r, cfg.tile_sample_min_width // r
    ts_h, ts_w = cfg.tile_sample_stride_height // r, cfg.tile_sample_stride_width // r
    blend_h = max(cfg.tile_sample_min_height - cfg.tile_sample_stride_height, 0)
    blend_w = max(cfg.tile_sample_min_width - cfg.tile_sample_stride_width, 0)
    rows = []
    for i in range(0, height, ts_h):
        row = []
        for j in range(0, width, ts_w):
            tile = z[:, :, :, i:min(i + tmin_h, height), j:min(j + tmin_w, width)]
            row.append(decoder_forward(p, cfg, tile, temb, dtype, noise))
        rows.append(row)
    prev: List[Tensor] = []
    result_rows = []
    for ri, row in enumerate(rows):
        cur: List[Tensor] = []
        out_row = []
        for cj, tile in enumerate(row):
            if ri > 0:
                tile = _blend(prev[cj], tile, blend_h, 3)
            if cj > 0:
                tile = _blend(cur[cj - 1], tile, blend_w, 4)
            cur.append(tile)
            hs = min(cfg.tile_sample_stride_height, tile.shape[3])
            ws = min(cfg.tile_sample_stride_width, tile.shape[4])
            out_row.append(tile[:, :, :, :hs, :ws])
        result_rows.append(torch.cat(out_row, 4))
        prev = cur
    dec = torch.cat(result_rows, 3)
    return dec[:, :, :, :height * r, :width * r]


def temporal_tiled_decode(p, cfg: VaeConfig, z: Tensor, temb, dtype, use_tiling: bool, noise: Optional[NoisePlanes] = None) -> Tensor:
    """AutoencoderKLLtxVideo::temporal_tiled_decode (vae.rs:2358-2434)."""
    nf = z.shape[2]
    tr, r = cfg.temporal_compression_ratio, cfg.spatial_compression_ratio
    num_sample_frames = (nf - 1) * tr + 1
    tmin_h, tmin_w = cfg.tile_sample_min_height // r, cfg.tile_sample_min_width // r
    tmin_t = cfg.tile_sample_min_num_frames // tr
    tstride_t = cfg.tile_sample_stride_num_frames // tr
    blend_t = max(cfg.tile_sample_min_num_frames - cfg.tile_sample_stride_num_frames, 0)
    row = []
    for li, i in enumerate(range(0, nf, tstride_t)):
        tile = z[:, :, i:min(i + tmin_t + 1, nf)]
        if use_tiling and (tile.shape[3] > tmin_h or tile.shape[4] > tmin_w):
            dec = tiled_decode(p, cfg, tile, temb, dtype, noise)
        else:
            dec = decoder_forward(p, cfg, tile, temb, dtype, noise)
        if li > 0 and dec.shape[2] > 1:
            dec = dec[:, :, :-1]
        row.append(dec)
    out = []
    for idx, tile in enumerate(row):
        if idx > 0:
            bl = _blend(row[idx - 1], tile, blend_t, 2)
            out.append(bl[:, :, :min(cfg.tile_sample_stride_num_frames, bl.shape[2])])
        else:
            out.append(tile[:, :, :min(cfg.tile_sample_stride_num_frames + 1, tile.shape[2])])
    return torch.cat(out, 2)[:, :, :num_sample_frames]


def vae_decode(p, cfg: VaeConfig, z: Tensor, temb: Optional[Tensor], dtype=torch.float32,
               use_tiling: bool = False, use_framewise_decoding: bool = False, noise: Optional[NoisePlanes] = None) -> Tensor:
    """AutoencoderKLLtxVideo::decode -> decode_z (vae.rs:2101-2136, 2037-2066)."""
    z = z.to(dtype)
    t = temb.to(dtype) if temb is not None else None
    _, _, tt, hh, ww = z.shape
    r, tr = cfg.spatial_compression_ratio, cfg.temporal_compression_ratio
    if use_framewise_decoding and tt > cfg.tile_sample_min_num_frames // tr:
        return temporal_tiled_decode(p, cfg, z, t, dtype, use_tiling, noise)
    if use_tiling and (ww > cfg.tile_sample_min_width // r or hh > cfg.tile_sample_min_height // r):
        return tiled_decode(p, cfg, z, t, dtype, noise)
    return decoder_forward(p, cfg, z, t, dtype, noise)


def vae_decoder_weight_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    """Weight names (relative to `decoder.`) of LtxVideoDecoder3d::new (vae.rs:1521-1608)."""
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(name, i, o):
        s[name + ".conv.weight"] = (o, i, 3, 3, 3)
        s[name + ".conv.bias"] = (o,)

    def temb(name, dim):
        s[name + ".timestep_embedder.linear_1.weight"] = (dim, 256)
        s[name + ".timestep_embedder.linear_1.bias"] = (dim,)
        s[name + ".timestep_embedder.linear_2.weight"] = (dim, dim)
        s[name + ".timestep_embedder.linear_2.bias"] = (dim,)

    def resnet(name, c, inject=False):
        conv(name + ".conv1", c, c)
        conv(name + ".conv2", c, c)
        if cfg.timestep_conditioning:
            s[name + ".scale_shift_table"] = (4, c)
        if inject:                                                 # vae.rs:676-689 (the name the reference looks up)
            s[name + ".per_channel_scale1.weight"] = (c, 1, 1)
            s[name + ".per_channel_scale2.weight"] = (c, 1, 1)

    mid = cfg.mid_channels
    nres = list(reversed(cfg.decoder_layers_per_block))
    upf = list(reversed(cfg.decoder_upsample_factor))
    inj = list(reversed(cfg.decoder_inject_noise))
    sts = list(reversed(cfg.decoder_spatiotemporal_scaling))
    conv("conv_in", cfg.latent_channels, mid)
    if cfg.timestep_conditioning:
        temb("mid_block.time_embedder", 4 * mid)
    for i in range(nres[0]):
        resnet(f"mid_block.resnets.{i}", mid, bool(inj[0]) if inj else False)
    cur = mid
    for bi, ch in enumerate(cfg.stage_channels()):
        pre = f"up_blocks.{bi}"
        conv(pre + ".upsamplers.0.conv", ch * upf[bi], ch * (8 if (bi >= len(sts) or sts[bi]) else 4))     # vae.rs:1063
        if cfg.timestep_conditioning:
            temb(pre + ".time_embedder", 4 * ch)
        for i in range(nres[bi + 1]):
            resnet(pre + f".resnets.{i}", ch, bool(inj[bi + 1]) if bi + 1 < len(inj) else False)
        cur = ch
    conv("conv_out", cur, cfg.out_channels * cfg.patch_size * cfg.patch_size)
    if cfg.timestep_conditioning:
        temb("time_embedder", 2 * cur)
        s["scale_shift_table"] = (2, cur)
        s["timestep_scale_multiplier"] = ()
    return s


# --------------------------------------------------------------------------
# Scheduler  (src/models/ltx_video/scheduler.rs)
# --------------------------------------------------------------------------

@dataclass
class SchedulerCfg:                   # configs.rs:101-121 common_scheduler_config
    num_train_timesteps: int = 1000
    shift: float = 1.0
    shift_terminal: Optional[float] = 0.1
    stochastic_sampling: bool = False
    use_karras_sigmas: bool = False   # scheduler.rs:30-32; at most one of the three (:85-93); no preset of configs.rs enables them
    use_exponential_sigmas: bool = False
    use_beta_sigmas: bool = False
    invert_sigmas: bool = False       # scheduler.rs:389-399


class FlowMatchEulerScheduler:
    """FlowMatchEulerDiscreteScheduler: exponential time shift, optional stretch-to-terminal, the karras / exponential /
    beta sigma conversions and invert_sigmas; scheduler.rs:84-146, 172-272, 274-441, 495-595, 646-668.  All scalar math in
    np.float32 like the Rust f32 code (convert_to_beta goes through f64, as the Rust does)."""

    def __init__(self, cfg: SchedulerCfg = SchedulerCfg()):
        if int(cfg.use_karras_sigmas) + int(cfg.use_exponential_sigmas) + int(cfg.use_beta_sigmas) > 1:
            raise ValueError("Only one of use_beta_sigmas/use_exponential_sigmas/use_karras_sigmas can be enabled.")   # :85-93
        self.cfg = cfg
        n = cfg.num_train_timesteps
        ts = np.arange(n, 0, -1, dtype=np.float32)
        sig = ts / np.float32(n)
        sh = np.float32(cfg.shift)
        sig = sh * sig / (np.float32(1.0) + (sh - np.float32(1.0)) * sig)       # :105-113
        self.sigma_min = np.float32(sig[-1])
        self.sigma_max = np.float32(sig[0])
        self.sigmas = np.concatenate([sig, np.zeros(1, np.float32)])
        self.timesteps = (sig * np.float32(n)).astype(np.float32)
        self.step_index: Optional[int] = None

    @staticmethod
    def _linspace(start, end, steps):                               # :209-220
        if steps == 0:
            return np.zeros(0, np.float32)
        if steps == 1:
            return np.array([start], np.float32)
        i = np.arange(steps, dtype=np.float32)
        return (np.float32(start) + (np.float32(end) - np.float32(start)) * i / np.float32(steps - 1)).astype(np.float32)

    def set_timesteps(self, num_inference_steps: Optional[int] = None, sigmas: Optional[Sequence[float]] = None,
                      mu: Optional[float] = None) -> List[int]:
        """set_timesteps (:274-412) + trait wrapper (:646-660): returns the
        timesteps truncated to i64."""
        n = self.cfg.num_train_timesteps
        if sigmas is not None:
            sig = np.asarray(sigmas, dtype=np.float32)
        else:
            tsv = self._linspace(self.sigma_max * np.float32(n), self.sigma_min * np.float32(n), num_inference_steps)
            sig = (tsv / np.float32(n)).astype(np.float32)
        if mu is not None:                                          # :341-346, time_shift_scalar :172-186
            emu = np.exp(np.float32(mu), dtype=np.float32)
            with np.errstate(divide="ignore"):
                base = np.power((np.float32(1.0) / sig - np.float32(1.0)).astype(np.float32), np.float32(1.0), dtype=np.float32)
            sig = (emu / (emu + base)).astype(np.float32)
        else:
            sh = np.float32(self.cfg.shift)
            sig = (sh * sig / (np.float32(1.0) + (sh - np.float32(1.0)) * sig)).astype(np.float32)
        if self.cfg.shift_terminal is not None and len(sig):       # :188-207
            one_minus_last = np.float32(1.0) - sig[-1]
            scale = one_minus_last / (np.float32(1.0) - np.float32(self.cfg.shift_terminal))
            sig = (np.float32(1.0) - (np.float32(1.0) - sig) / scale).astype(np.float32)
        nsteps = len(sig)
        if self.cfg.use_karras_sigmas:                              # convert_to_karras :222-235 (rho = 7)
            smin, smax = np.float32(sig[-1]), np.float32(sig[0])
            rho = np.float32(7.0)
            ramp = self._linspace(0.0, 1.0, nsteps)
            mn, mx = np.power(smin, np.float32(1.0) / rho, dtype=np.float32), np.power(smax, np.float32(1.0) / rho, dtype=np.float32)
            sig = np.power((mx + ramp * (mn - mx)).astype(np.float32), rho, dtype=np.float32)
        elif self.cfg.use_exponential_sigmas:                       # convert_to_exponential :237-245
            smin, smax = np.float32(sig[-1]), np.float32(sig[0])
            with np.errstate(divide="ignore"):
                sig = np.exp(self._linspace(np.log(smax, dtype=np.float32), np.log(smin, dtype=np.float32), nsteps), dtype=np.float32)
        elif self.cfg.use_beta_sigmas:                              # convert_to_beta :247-272 (alpha = beta = 0.6, statrs inverse_cdf = scipy ppf)
            from scipy.stats import beta as _beta
            smin, smax = np.float32(sig[-1]), np.float32(sig[0])
            ts = 1.0 - self._linspace(0.0, 1.0, nsteps).astype(np.float64)
            ppf = _beta.ppf(ts, 0.6, 0.6)
            sig = (float(smin) + ppf * float(np.float32(smax - smin))).astype(np.float32)
        self.timesteps = (sig * np.float32(n)).astype(np.float32)
        if self.cfg.invert_sigmas:                                  # :389-399
            sig = (np.float32(1.0) - sig).astype(np.float32)
            self.timesteps = (sig * np.float32(n)).astype(np.float32)
            self.sigmas = np.concatenate([sig, np.ones(1, np.float32)]).astype(np.float32)
        else:
            self.sigmas = np.concatenate([sig, np.zeros(1, np.float32)]).astype(np.float32)
        self.step_index = None
        return [int(x) for x in self.timesteps]                    # `as i64` truncation (:659)

    def step(self, model_output: Tensor, timestep: float, sample: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        """step (:495-595), non per-token path; result stays f32."""
        if self.step_index is None:                                 # init_step_index :433-440
            idx = [i for i, v in enumerate(self.timesteps) if abs(float(v) - float(timestep)) < 1e-6]
            if not idx:
                raise ValueError("timestep not found in schedule_timesteps.")
            self.step_index = idx[1] if len(idx) > 1 else idx[0]
        s = sample.float()
        sigma = np.float32(self.sigmas[self.step_index])
        sigma_next = np.float32(self.sigmas[self.step_index + 1])
        if self.cfg.stochastic_sampling:
            x0 = s - float(sigma) * model_output.float()
            if noise is None:
                noise = torch.randn_like(s)
            out = (1.0 - float(sigma_next)) * x0 + float(sigma_next) * noise
        else:
            dt = np.float32(sigma_next - sigma)
            out = s + model_output.float() * float(dt)
        self.step_index += 1
        return out


# --------------------------------------------------------------------------
# Pipeline  (src/models/ltx_video/t2v_pipeline.rs)
# --------------------------------------------------------------------------

def calculate_shift(seq_len: int, base_seq_len: int = 256, max_seq_len: int = 4096,
                    base_shift: float = 0.5, max_shift: float = 1.15) -> float:
    """t2v_pipeline.rs:159-169 in f32."""
    m = (np.float32(max_shift) - np.float32(base_shift)) / np.float32(max_seq_len - base_seq_len)
    b = np.float32(base_shift) - m * np.float32(base_seq_len)
    return float(np.float32(seq_len) * m + b)


def pack_latents(x: Tensor, p: int = 1, pt: int = 1) -> Tensor:
    """t2v_pipeline.rs:474-504."""
    b, c, f, h, w = x.shape
    x = x.reshape(b, c, f // pt, pt, h // p, p, w // p, p).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return x.flatten(4).reshape(b, (f // pt) * (h // p) * (w // p), -1)


def unpack_latents(x: Tensor, f: int, h: int, w: int, p: int = 1, pt: int = 1) -> Tensor:
    """t2v_pipeline.rs:506-550."""
    b, _, d = x.shape
    c = d // (pt * p * p)
    x = x.reshape(b, f, h, w, c, pt, p, p).permute(0, 4, 1, 5, 2, 6, 3, 7).contiguous()
    return x.reshape(b, c, f * pt, h * p, w * p)


def normalize_latents(x, mean, std, sf: float = 1.0):
    """t2v_pipeline.rs:552-571."""
    c = x.shape[1]
    return (x - mean.reshape(1, c, 1, 1, 1).to(x.dtype)) * sf / std.reshape(1, c, 1, 1, 1).to(x.dtype)


def denormalize_latents(x, mean, std, sf: float = 1.0):
    """t2v_pipeline.rs:573-594."""
    c = x.shape[1]
    return x * std.reshape(1, c, 1, 1, 1).to(x.dtype) * (1.0 / sf) + mean.reshape(1, c, 1, 1, 1).to(x.dtype)


def build_video_coords(batch: int, f: int, h: int, w: int, frame_rate: int = 25,
                       ts_ratio: int = 8, sp_ratio: int = 32) -> Tensor:
    """t2v_pipeline.rs:798-847: [B, S, 3] f32."""
    gf = torch.arange(f, dtype=torch.float32).reshape(f, 1, 1).expand(f, h, w)
    gh = torch.arange(h, dtype=torch.float32).reshape(1, h, 1).expand(f, h, w)
    gw = torch.arange(w, dtype=torch.float32).reshape(1, 1, w).expand(f, h, w)
    vc = torch.stack([gf, gh, gw], 0).flatten(1).transpose(0, 1).unsqueeze(0)
    vf = (vc[..., 0] * float(ts_ratio) + (1.0 - float(ts_ratio))).clamp(0.0, 1000.0) * (1.0 / frame_rate)
    vh = vc[..., 1] * float(sp_ratio)
    vw = vc[..., 2] * float(sp_ratio)
    return torch.stack([vf, vh, vw], -1).expand(batch, f * h * w, 3).contiguous()


def std_except0(x: Tensor) -> Tensor:
    """std_over_dims_except0_keepdim (t2v_pipeline.rs:209-224), unbiased."""
    b = x.shape[0]
    return x.flatten(1).var(1, unbiased=True, keepdim=True).sqrt().reshape([b] + [1] * (x.dim() - 1))


def rescale_noise_cfg(noise_cfg: Tensor, noise_text: Tensor, guidance_rescale: float) -> Tensor:
    """t2v_pipeline.rs:227-243."""
    ratio = std_except0(noise_text) / std_except0(noise_cfg)
    resc = noise_cfg * ratio
    g = float(np.float32(guidance_rescale))
    return resc * g + noise_cfg * float(np.float32(1.0) - np.float32(guidance_rescale))


def guidance_combine(text: Tensor, uncond: Optional[Tensor], perturbed: Optional[Tensor],
                     guidance_scale: float, guidance_rescale: float, stg_scale: float) -> Tensor:
    """t2v_pipeline.rs:941-964 (all f32)."""
    text = text.float()
    combined = text.clone()
    if uncond is not None:
        u = uncond.float()
        combined = u + (text - u) * float(np.float32(guidance_scale))
        if guidance_rescale > 0.0:
            combined = rescale_noise_cfg(combined, text, guidance_rescale)
    if perturbed is not None:
        combined = combined + (text - perturbed.float()) * float(np.float32(stg_scale))
    return combined


def postprocess_video(v: Tensor) -> Tensor:
    """LtxVideoProcessor::postprocess_video (t2v_pipeline.rs:146-155)."""
    return (v * 0.5 + 0.5).clamp(0.0, 1.0) * 255.0


@dataclass
class PipelineArgs:
    height: int
    width: int
    num_frames: int
    frame_rate: int = 25
    num_inference_steps: int = 7
    sigmas: Optional[List[float]] = None
    guidance_scale: float = 1.0
    guidance_rescale: float = 0.0
    stg_scale: float = 0.0
    skip_block_list: Optional[List[int]] = None
    decode_timestep: float = 0.05
    decode_noise_scale: Optional[float] = 0.025
    output_latent: bool = False
    use_tiling: bool = False
    use_framewise_decoding: bool = False


def pipeline_call(dit_p, dit_cfg: DitConfig, vae_p, vae_cfg: VaeConfig, latents_mean: Tensor, latents_std: Tensor,
                  args: PipelineArgs, latents: Tensor, prompt_embeds: Tensor, prompt_mask: Tensor,
                  neg_embeds: Optional[Tensor] = None, neg_mask: Optional[Tensor] = None,
                  decode_noise: Optional[Tensor] = None, dtype=torch.float32,
                  sched_cfg: SchedulerCfg = SchedulerCfg(), trajectory: Optional[list] = None,
                  step_noise: Optional[Tensor] = None, timestep_cast: Optional[torch.dtype] = None,
                  interrupt_at: Optional[int] = None) -> Tensor:
    """LtxPipeline::call (t2v_pipeline.rs:627-1073) with embeddings supplied
    (text encoder out of scope) and the decode noise supplied explicitly (the
    reference draws it from the device RNG, :1055).  interrupt_at = k: `self.interrupt`
    raised before step k - that and every later step is skipped (`continue`, :861-863),
    the decode of the latents reached so far still runs."""
    do_cfg = args.guidance_scale > 1.0
    do_stg = args.stg_scale > 0.0
    skip_perm: Sequence[int] = ()
    if args.skip_block_list is not None and not do_stg:            # :691-697
        skip_perm = list(args.skip_block_list)
    lat = latents.float()
    F_ = (args.num_frames - 1) // vae_cfg.temporal_compression_ratio + 1
    H_ = args.height // vae_cfg.spatial_compression_ratio
    W_ = args.width // vae_cfg.spatial_compression_ratio
    S = F_ * H_ * W_
    has_custom = args.sigmas is not None
    sig = list(args.sigmas) if has_custom else list(FlowMatchEulerScheduler._linspace(1.0, 1.0 / args.num_inference_steps, args.num_inference_steps))
    mu = 0.0 if has_custom else calculate_shift(S)
    sched = FlowMatchEulerScheduler(sched_cfg)
    ts = sched.set_timesteps(sigmas=sig, mu=mu)
    b = lat.shape[0]
    coords = build_video_coords(b, F_, H_, W_, args.frame_rate, vae_cfg.temporal_compression_ratio, vae_cfg.spatial_compression_ratio)
    L = dit_cfg.num_layers

    def fwd(emb, mask, t, slm=None):
        tt = torch.full((b,), float(t))
        if timestep_cast is not None:     # test aid: the timestep as a `timestep_cast` model sees it (:1051), all else in `dtype`
            tt = tt.to(timestep_cast).float()
        return dit_forward(dit_p, dit_cfg, lat, emb, tt, mask, F_, H_, W_,
                           None, coords, slm, skip_perm, dtype)

    trajectory_idx: list = []
    for i_step, t in enumerate(ts):
        if interrupt_at is not None and i_step >= interrupt_at:
            continue
        if do_cfg or do_stg:
            un = fwd(neg_embeds, neg_mask, t) if do_cfg else None
            tx = fwd(prompt_embeds, prompt_mask, t)
            pe = None
            if do_stg:
                m = torch.zeros(L, b)
                for li in (args.skip_block_list or []):
                    if li < L:
                        m[li] = 1.0
                pe = fwd(prompt_embeds, prompt_mask, t, m)
            noise_pred = guidance_combine(tx, un, pe, args.guidance_scale, args.guidance_rescale, args.stg_scale)
        else:
            noise_pred = fwd(prompt_embeds, prompt_mask, t).float()
        # stochastic sampling draws randn_like(sample) per step (scheduler.rs:567); supplied explicitly here
        lat = sched.step(noise_pred, float(t), lat, None if step_noise is None else step_noise[len(trajectory_idx)])
        trajectory_idx.append(0)
        if trajectory is not None:
            trajectory.append(lat.clone())
    if args.output_latent:
        return lat
    x = unpack_latents(lat, F_, H_, W_)
    x = denormalize_latents(x, latents_mean, latents_std, vae_cfg.scaling_factor)
    temb = None
    if vae_cfg.timestep_conditioning:
        temb = torch.full((b,), args.decode_timestep, dtype=torch.float32)
        sc = args.decode_timestep if args.decode_noise_scale is None else args.decode_noise_scale
        if decode_noise is not None:      # caller-supplied noise; None = skip the mix (API extension)
            x = x * (1.0 - sc) + decode_noise.to(x.dtype) * sc
    x = x.to(dtype)
    v = vae_decode(vae_p, vae_cfg, x, temb, dtype, args.use_tiling, args.use_framewise_decoding)
    return postprocess_video(v)


# --------------------------------------------------------------------------
# Synthetic weights (bench / tests; real checkpoints are unavailable offline)
# --------------------------------------------------------------------------

def _name_seed(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & _M64
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def synth_weights(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Deterministic per-tensor-name synthetic init that keeps activations
    O(1) through the network (so attention logits are not degenerate):
    matmul/conv weights ~ N(0,1)/sqrt(fan_in); biases ~ 0.02 N; norm weights
    ~ 1 + 0.1 N; scale_shift_table ~ N/sqrt(dim) (the torch init quoted at
    ltx_transformer.rs:806); timestep_scale_multiplier = 1000."""
    out = {}
    for name, shp in shapes.items():
        g = torch.Generator().manual_seed(_name_seed(name, seed))
        if name.endswith("timestep_scale_multiplier"):
            w = torch.tensor(1000.0)
        elif "norm_q" in name or "norm_k" in name:
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith("scale_shift_table"):
            w = torch.randn(shp, generator=g) / math.sqrt(shp[-1])
        elif name.endswith(".bias"):
            w = 0.02 * torch.randn(shp, generator=g)
        else:
            fan_in = int(np.prod(shp[1:]))
            w = torch.randn(shp, generator=g) / math.sqrt(fan_in)
        out[name] = w.to(dtype)
    return out


# --------------------------------------------------------------------------
# T5 v1.1 encoder (the step before the path)
# --------------------------------------------------------------------------
# The reference wraps `candle_transformers::models::t5::T5EncoderModel` (text_encoder.rs:315-345, 597-606; crates.io
# candle-transformers ^0.9.2, absent from the checkout) configured by `to_candle_t5_config` (:222-249): gated NewGelu
# feed-forward, bidirectional relative-position bias with 32 buckets / max distance 128, T5LayerNorm, no attention mask
# (`VTextEncoder::forward(input_ids)` passes ids only, :600-604).  That model is a port of Hugging Face's T5: this
# restatement follows the published algorithm and is PINNED against `transformers.T5EncoderModel` (installed here) on
# shared random weights in tests/test_t5_cpu.py.

@dataclass
class T5Config:                       # text_encoder.rs:66-113, presets :169-203
    vocab_size: int = 32128
    d_model: int = 4096
    d_kv: int = 64
    d_ff: int = 10240
    num_layers: int = 24
    num_heads: int = 64
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6


def t5_weight_shapes(cfg: T5Config) -> Dict[str, Tuple[int, ...]]:
    inner = cfg.num_heads * cfg.d_kv
    s: Dict[str, Tuple[int, ...]] = {"shared.weight": (cfg.vocab_size, cfg.d_model)}
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}.layer."
        for n in "qkv":
            s[p + f"0.SelfAttention.{n}.weight"] = (inner, cfg.d_model)
        s[p + "0.SelfAttention.o.weight"] = (cfg.d_model, inner)
        if i == 0:
            s[p + "0.SelfAttention.relative_attention_bias.weight"] = (cfg.relative_attention_num_buckets, cfg.num_heads)
        s[p + "0.layer_norm.weight"] = (cfg.d_model,)
        s[p + "1.DenseReluDense.wi_0.weight"] = (cfg.d_ff, cfg.d_model)
        s[p + "1.DenseReluDense.wi_1.weight"] = (cfg.d_ff, cfg.d_model)
        s[p + "1.DenseReluDense.wo.weight"] = (cfg.d_model, cfg.d_ff)
        s[p + "1.layer_norm.weight"] = (cfg.d_model,)
    s["encoder.final_layer_norm.weight"] = (cfg.d_model,)
    return s


def t5_relative_position_bucket(rel: Tensor, num_buckets: int, max_distance: int) -> Tensor:
    """Bidirectional bucketing (T5Attention._relative_position_bucket): rel = key_pos - query_pos."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float().clamp(min=1) / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(is_small, n, large)


def t5_position_bias(table: Tensor, S: int, num_buckets: int, max_distance: int) -> Tensor:
    """[H, S, S] additive bias from relative_attention_bias.weight [buckets, H] (computed once, shared by all layers)."""
    ctx = torch.arange(S)[:, None]; mem = torch.arange(S)[None, :]
    bucket = t5_relative_position_bucket(mem - ctx, num_buckets, max_distance)
    return table.float()[bucket].permute(2, 0, 1)


def t5_layer_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """T5LayerNorm: RMS without mean subtraction, variance in f32."""
    v = x.float().pow(2).mean(-1, keepdim=True)
    return (x.float() * torch.rsqrt(v + eps)).to(x.dtype) * w.to(x.dtype)


def t5_encoder_forward(p: Dict[str, Tensor], cfg: T5Config, input_ids: Tensor, dtype=torch.float32,
                       attention_mask: Optional[Tensor] = None) -> Tensor:
    """T5EncoderModel.forward(input_ids) -> last hidden state [B, S, d_model]; no attention mask (the bf16 wrapper's
    behaviour, text_encoder.rs:600-605).  attention_mask [B, S] (1 keep, 0 pad): QuantizedT5EncoderModel::forward's extended
    mask (quantized_t5_encoder.rs:624-634): (scores + position_bias) + (1 - mask) * -1e9 over the keys (:218-220)."""
    B, S = input_ids.shape
    mask_bias = None if attention_mask is None else ((1.0 - attention_mask.float()) * -1e9).reshape(B, 1, 1, S)
    H, dk = cfg.num_heads, cfg.d_kv
    w = lambda k: p[k].to(dtype)
    h = w("shared.weight")[input_ids]
    bias = t5_position_bias(p["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], S,
                            cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)
    for i in range(cfg.num_layers):
        pre = f"encoder.block.{i}.layer."
        n = t5_layer_norm(h, w(pre + "0.layer_norm.weight"), cfg.layer_norm_epsilon)
        q = (n @ w(pre + "0.SelfAttention.q.weight").T).reshape(B, S, H, dk).transpose(1, 2)
        k = (n @ w(pre + "0.SelfAttention.k.weight").T).reshape(B, S, H, dk).transpose(1, 2)
        v = (n @ w(pre + "0.SelfAttention.v.weight").T).reshape(B, S, H, dk).transpose(1, 2)
        sc = q.float() @ k.float().transpose(-1, -2) + bias[None]            # T5 does not scale by 1/sqrt(d_kv)
        if mask_bias is not None: sc = sc + mask_bias
        a = torch.softmax(sc, -1).to(dtype) @ v
        h = h + a.transpose(1, 2).reshape(B, S, H * dk) @ w(pre + "0.SelfAttention.o.weight").T
        n = t5_layer_norm(h, w(pre + "1.layer_norm.weight"), cfg.layer_norm_epsilon)
        g = gelu_approximate(n @ w(pre + "1.DenseReluDense.wi_0.weight").T) * (n @ w(pre + "1.DenseReluDense.wi_1.weight").T)
        h = h + g @ w(pre + "1.DenseReluDense.wo.weight").T
    return t5_layer_norm(h, w("encoder.final_layer_norm.weight"), cfg.layer_norm_epsilon)
