"""GPU suite: HIP kernels checked DIRECTLY against vectors produced by the reference's own runnable scripts
(tests/golden/ref_*.safetensors, made by tools/gen_fixtures.py from /root/reference/scripts executed unmodified) -
not only the oracle.  f32 mode, bars are the scripts' own (exact for index/integer work, 1e-5 / 1e-6)."""
import pytest
import torch

import ltx_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


def test_unpatchify_epilogue_vs_reference_script(hip, golden):
    """scripts/test_unpatchify.py: x[c,t,h,w] = 1000c + 100t + 10h + w through the decoder's permute(0,1,5,2,6,4,7,3)
    (vae.rs:1626-1654).  The HIP path has no standalone unpatchify: it is the conv_out epilogue (EPI_UNPATCH), so the
    tensor goes through a 3x3x3 conv whose only non-zero tap is the centre identity (exact in f32)."""
    u = golden("ref_unpatchify.safetensors")
    x = u["x"]                                               # [1,48,2,4,6]
    w = torch.zeros(48, 48, 3, 3, 3); w[torch.arange(48), torch.arange(48), 1, 1, 1] = 1.0
    got = hip.ops.conv_out_unpatchify(x.permute(0, 2, 3, 4, 1).contiguous().to(DEV), w.to(DEV), torch.zeros(48).to(DEV))
    assert got.shape == u["out"].shape and torch.equal(got.cpu(), u["out"])


def test_denormalize_kernel_vs_reference_script(hip, golden):
    """scripts/gen_latent_norm_ref.py (t2v_pipeline.rs:573-594): normalized -> denormalized through ltx_vae_prepare_latents
    (the denormalize + noise-mix kernel of the decode entry), with the script's mean / std / scaling factor."""
    n = golden("ref_latent_norm.safetensors")
    z, want = n["normalized"], n["denormalized"]           # [1,128,3,4,6]
    B, Cc, F, H, W = z.shape
    vcfg = dict(latent_channels=Cc, decoder_block_out_channels=(16, 32, 64), decoder_layers_per_block=(1, 1, 1, 1), scaling_factor=float(n["scaling_factor"]))
    ocfg = O.VaeConfig(**vcfg)
    vw = {"decoder." + k: v.to(DEV) for k, v in O.synth_weights(O.vae_decoder_weight_shapes(ocfg), seed=3).items()}
    vw["latents_mean"] = n["latents_mean"].to(DEV); vw["latents_std"] = n["latents_std"].to(DEV)
    vae = hip.AutoencoderKLLtxVideo(hip.AutoencoderKLLtxVideoConfig(**vcfg), vw, torch.float32)
    got = vae.prepare_latents(O.pack_latents(z).to(DEV), F, H, W)
    assert (got.cpu() - want).abs().max() < 1e-5
    # and the noise mix on top (t2v_pipeline.rs:1049-1062): (1 - s) x + s eps
    eps = torch.randn(z.shape, generator=torch.Generator().manual_seed(2))
    got2 = vae.prepare_latents(O.pack_latents(z).to(DEV), F, H, W, eps.to(DEV), [0.025])
    assert (got2.cpu() - (want * 0.975 + eps * 0.025)).abs().max() < 1e-5


def test_rotary_kernel_vs_reference_script(hip, golden):
    """scripts/test_rope_rotation.py rust_apply_rotary_emb_linear.  The HIP kernel fuses q/k RMSNorm with the rotation;
    with unit weights the norm is a per-row scale and the rotation is linear, so kernel(x) * rms(x) must equal the
    script's rotation of x (tables: one value per channel pair, as the model builds them)."""
    r = golden("ref_rope_rotation.safetensors")
    x, cos, sin, want = r["x"][0], r["cos"][0], r["sin"][0], r["out"][0]
    assert torch.equal(cos[:, ::2], cos[:, 1::2]) and torch.equal(sin[:, ::2], sin[:, 1::2])
    eps = 1e-12
    y = hip.ops.qknorm_rope(x.to(DEV), torch.ones(x.shape[1]).to(DEV), eps, cos[:, ::2].contiguous().to(DEV), sin[:, ::2].contiguous().to(DEV)).cpu()
    rms = (x.pow(2).mean(-1, keepdim=True) + eps).sqrt()
    assert (y * rms - want).abs().max() < 1e-5, (y * rms - want).abs().max()


def test_pcg32_randn_vs_reference_script(hip, golden):
    r = golden("ref_rng.safetensors")
    assert (hip.pcg32_randn(42, (257,)) - r["randn"]).abs().max() < 1e-6          # verify_rng.py's own bar
    assert torch.equal(hip.pcg32_u32(42, 64), r["u32"])


def test_default_output_gif_from_device_video(hip, tmp_path):
    """main.rs:653-707 in one call (ltx_save_video_gif): device f32 video -> RGB8 -> GIF with the reference's settings;
    decoded back by the independent reader of tests/test_frames_cpu.py (64 x 96 frames: 205 training samples at speed 30)."""
    from test_frames_cpu import psnr_u8, read_gif
    B, F, H, W = 1, 4, 64, 96
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    base = torch.stack([80 + 100 * xx / W, 60 + 120 * yy / H, 200 - 90 * xx / W])          # [3,H,W], a narrow colour sheet
    v = torch.stack([base * (0.6 + 0.1 * f) for f in range(F)], 1)[None] + 0.4             # [1,3,F,H,W], fractional values
    v[0, :, 0, :4, :4] = 300.0; v[0, :, 1, :4, :4] = -20.0                                   # out of range: clamped
    path = str(tmp_path / "video.gif")
    hip.save_video_gif(v.to(DEV), path)
    w, h, loop, dec = read_gif(path)
    assert (w, h, loop, len(dec)) == (W, H, 0, B * F)
    want = v.permute(0, 2, 3, 4, 1).clamp(0, 255).to(torch.uint8)[0]
    for f in range(F):
        assert dec[f][0] == 4 and psnr_u8(dec[f][1], want[f]) > 30.0, (f, psnr_u8(dec[f][1], want[f]))


@pytest.mark.parametrize("dim", [2048, 4096])
def test_rope_table_kernel_vs_reference_scripts(hip, golden, dim):
    """rope_table_kernel (the DiT's per-forward table, half-width: one value per channel pair) against the reference's own
    scripts/compare_rope_freqs.py rust_compute_freqs / diffusers_compute_freqs and scripts/debug_rope.py
    prepare_video_coords_debug, executed unmodified by tools/gen_fixtures.py rope.  Bar: tests/verify_rope_parity.rs:253-254
    (MSE < 1e-5); left pad of dim % 6 columns exact; small angles to 1e-5.  dim 2048 = the 2B model, 4096 = the 13B."""
    r = golden("ref_rope_table.safetensors")
    F_, H_, W_ = (int(v) for v in r["fhw"])
    scale = tuple(float(v) for v in r["scale"])
    pad = (dim % 6) // 2
    mse = lambda a, b: float(((a.double() - b.double()) ** 2).mean())
    # grid path (ltx_transformer.rs:373-433), batch 2: the script's grid has two identical batch rows
    c, s = hip.ops.rope_table(2, F_, H_, W_, dim, rope_scale=scale)
    S = F_ * H_ * W_
    for b in range(2):
        cb, sb = c.cpu()[b * S:(b + 1) * S], s.cpu()[b * S:(b + 1) * S]
        wc, ws = r[f"rust_cos_{dim}_scaled"][0], r[f"rust_sin_{dim}_scaled"][0]
        assert torch.equal(wc[:, 0::2], wc[:, 1::2])                                  # the table IS pairs: half-width loses nothing
        assert mse(cb, wc[:, 0::2]) < 1e-5 and mse(sb, ws[:, 0::2]) < 1e-5
        assert mse(cb, r[f"diffusers_cos_{dim}_scaled_even"][0]) < 1e-5 and mse(sb, r[f"diffusers_sin_{dim}_scaled_even"][0]) < 1e-5
        assert torch.equal(cb[:, :pad], torch.ones(S, pad)) and torch.equal(sb[:, :pad], torch.zeros(S, pad))
        assert (cb[:, pad:pad + 48] - wc[:, 2 * pad:2 * pad + 96:2]).abs().max() < 1e-5
    # video_coords path (:449-463): coords = fractional grid * base
    grid = r["grid_rand"]
    coords = (grid * torch.tensor([20.0, 2048.0, 2048.0]))[0]
    n = coords.shape[0]
    c2, s2 = hip.ops.rope_table(1, 1, 1, n, dim, coords=coords.to(DEV))
    assert mse(c2.cpu(), r[f"rust_cos_{dim}_rand"][0][:, 0::2]) < 1e-5 and mse(s2.cpu(), r[f"rust_sin_{dim}_rand"][0][:, 0::2]) < 1e-5
    assert (c2.cpu()[:, pad:pad + 48] - r[f"rust_cos_{dim}_rand"][0][:, 2 * pad:2 * pad + 96:2]).abs().max() < 1e-5
