"""Weight names and shapes the two models are built from - the constructor contract of the reference
(`LtxVideoTransformer3DModel::new`, ltx_transformer.rs:957-1003, 570-577, 777-808, 160-161;
`LtxVideoDecoder3d::new`, vae.rs:1521-1608, 323-335, 1827-1838), as the library's loaders (csrc/dit.hip `build`,
csrc/vae.hip) consume it.  Used by bench.py and the tests to synthesise random-init weights of the real architecture
(no checkpoints offline); linear weights are [out, in], conv weights [O, I, 3, 3, 3]."""
from __future__ import annotations

from typing import Dict, Tuple

Shape = Tuple[int, ...]


def dit_weight_shapes(cfg) -> Dict[str, Shape]:
    """cfg: LtxVideoTransformer3DModelConfig (any object with its fields)."""
    D = cfg.num_attention_heads * cfg.attention_head_dim
    s: Dict[str, Shape] = {}

    def linear(name: str, fan_in: int, fan_out: int):
        s[name + ".weight"] = (fan_out, fan_in)
        s[name + ".bias"] = (fan_out,)

    linear("proj_in", cfg.in_channels, D)
    s["scale_shift_table"] = (2, D)
    linear("time_embed.emb.timestep_embedder.linear_1", 256, D)
    linear("time_embed.emb.timestep_embedder.linear_2", D, D)
    linear("time_embed.linear", D, 6 * D)
    linear("caption_projection.linear_1", cfg.caption_channels, D)
    linear("caption_projection.linear_2", D, D)
    for i in range(cfg.num_layers):
        blk = f"transformer_blocks.{i}."
        for attn, kv_dim in (("attn1", D), ("attn2", cfg.cross_attention_dim)):
            linear(blk + attn + ".to_q", D, D)
            linear(blk + attn + ".to_k", kv_dim, D)
            linear(blk + attn + ".to_v", kv_dim, D)
            linear(blk + attn + ".to_out.0", D, D)
            s[blk + attn + ".norm_q.weight"] = (D,)
            s[blk + attn + ".norm_k.weight"] = (D,)
        linear(blk + "ff.net.0.proj", D, 4 * D)
        linear(blk + "ff.net.2", 4 * D, D)
        s[blk + "scale_shift_table"] = (6, D)
    linear("proj_out", D, cfg.out_channels)
    return s


def vae_decoder_weight_shapes(cfg) -> Dict[str, Shape]:
    """cfg: AutoencoderKLLtxVideoConfig; names are relative to `decoder.`."""
    s: Dict[str, Shape] = {}
    tc = cfg.timestep_conditioning

    def conv(name: str, cin: int, cout: int):
        s[name + ".conv.weight"] = (cout, cin, 3, 3, 3)
        s[name + ".conv.bias"] = (cout,)

    def time_embedder(name: str, dim: int):
        s[name + ".timestep_embedder.linear_1.weight"] = (dim, 256)
        s[name + ".timestep_embedder.linear_1.bias"] = (dim,)
        s[name + ".timestep_embedder.linear_2.weight"] = (dim, dim)
        s[name + ".timestep_embedder.linear_2.bias"] = (dim,)

    def resnet(name: str, c: int):
        conv(name + ".conv1", c, c)
        conv(name + ".conv2", c, c)
        if tc:
            s[name + ".scale_shift_table"] = (4, c)

    chans = list(reversed(cfg.decoder_block_out_channels))       # decoder runs widest first (vae.rs:1538-1545)
    layers = list(reversed(cfg.decoder_layers_per_block))
    upf = list(reversed(cfg.decoder_upsample_factor))
    mid = chans[0]
    conv("conv_in", cfg.latent_channels, mid)
    if tc:
        time_embedder("mid_block.time_embedder", 4 * mid)
    for i in range(layers[0]):
        resnet(f"mid_block.resnets.{i}", mid)
    cur = mid
    for bi in range(len(chans)):
        ch = cur // upf[bi]                                       # each up block halves the width (upsample factor 2)
        up = f"up_blocks.{bi}"
        conv(up + ".upsamplers.0.conv", cur, ch * 8)              # 8 = 2x2x2 depth-to-space
        if tc:
            time_embedder(up + ".time_embedder", 4 * ch)
        for i in range(layers[bi + 1]):
            resnet(up + f".resnets.{i}", ch)
        cur = ch
    conv("conv_out", cur, cfg.out_channels * cfg.patch_size * cfg.patch_size)
    if tc:
        time_embedder("time_embedder", 2 * cur)
        s["scale_shift_table"] = (2, cur)
        s["timestep_scale_multiplier"] = ()
    return s
