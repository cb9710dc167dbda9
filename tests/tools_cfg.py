"""Shared tiny-model configs (must match tools/gen_fixtures.py)."""
VAE_CFG = dict(latent_channels=8, decoder_block_out_channels=(32, 64, 128), decoder_layers_per_block=(1, 1, 1, 2))
PIPE_DIT_CFG = dict(in_channels=8, out_channels=8, num_attention_heads=2, attention_head_dim=16, cross_attention_dim=32,
                    num_layers=3, caption_channels=32)


def to_official_names(dit_names, vae_names):
    """Inverse of KeyRemapper for the synthetic checkpoints of the tests: diffusers-layout names -> the Official unified
    checkpoint's names ("model.diffusion_model." / "vae." prefixes, flat VAE block indices, native module names)."""
    import re
    out = {}
    for n in dit_names:
        o = n.replace("proj_in", "patchify_proj").replace("time_embed", "adaln_single").replace("norm_q", "q_norm").replace("norm_k", "k_norm")
        out["model.diffusion_model." + o] = ("dit", n)
    for n in vae_names:
        o = n
        if o.startswith("decoder.time_embedder"):
            o = o.replace("decoder.time_embedder", "decoder.last_time_embedder", 1)
        if o.startswith("decoder.scale_shift_table"):
            o = o.replace("decoder.scale_shift_table", "decoder.last_scale_shift_table", 1)
        o = o.replace("decoder.mid_block", "decoder.up_blocks.0")
        o = re.sub(r"decoder\.up_blocks\.(\d+)\.upsamplers\.0", lambda m: f"decoder.UP.{2 * int(m.group(1)) + 1}", o)
        o = re.sub(r"decoder\.up_blocks\.(\d+)(?=\.resnets|\.time_embedder|\.scale_shift|\.conv)", lambda m: f"decoder.UP.{2 * int(m.group(1)) + 2}", o) \
            if not n.startswith("decoder.mid_block") else o
        o = o.replace("decoder.UP.", "decoder.up_blocks.").replace("resnets", "res_blocks")
        o = o.replace("latents_mean", "per_channel_statistics.mean-of-means").replace("latents_std", "per_channel_statistics.std-of-means")
        out["vae." + o] = ("vae", n)
    return out
