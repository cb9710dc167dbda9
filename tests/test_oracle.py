"""CPU suite, part 1: the oracle is pinned against (a) vectors produced by the reference's own
runnable scripts, (b) the reference's closed-form tests, (c) independent torch library ops, and
(d) the committed oracle fixtures (drift guard).  No GPU, no /root/reference at run time."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ltx_oracle as O
from conftest import rel_max


# ---------- (a) reference-script vectors ----------
def test_guidance_matches_reference_script(golden):
    g = golden("ref_guidance.safetensors")          # scripts/gen_guidance_ref.py; bar: tests/verify_guidance_parity.rs:58-67
    cfg = O.guidance_combine(g["noise_pred_text"], g["noise_pred_uncond"], None, float(g["guidance_scale"]), 0.0, 0.0)
    assert (cfg - g["combined_cfg"]).abs().max() < 1e-5
    fin = O.guidance_combine(g["noise_pred_text"], g["noise_pred_uncond"], g["noise_pred_perturb"],
                             float(g["guidance_scale"]), 0.0, float(g["stg_scale"]))
    assert (fin - g["combined_final"]).abs().max() < 1e-5


def test_latent_norm_matches_reference_script(golden):
    n = golden("ref_latent_norm.safetensors")       # scripts/gen_latent_norm_ref.py; bar: tests/verify_latent_norm_parity.rs:65-79
    sf = float(n["scaling_factor"])
    assert (O.normalize_latents(n["latents"], n["latents_mean"], n["latents_std"], sf) - n["normalized"]).abs().max() < 1e-5
    assert (O.denormalize_latents(n["normalized"], n["latents_mean"], n["latents_std"], sf) - n["denormalized"]).abs().max() < 1e-5


# ---------- (b) the reference's closed-form tests ----------
def test_adaln_modulation_known_answer():
    # tests/verify_rope_parity.rs:646-733: x[0,0,1]=0.1, scale=0.01, shift=0.001 -> 0.102
    x = (torch.arange(2 * 4 * 8, dtype=torch.float32) * 0.1).reshape(2, 4, 8)
    scale = (torch.arange(2 * 8, dtype=torch.float32) * 0.01).reshape(2, 1, 8)
    shift = (torch.arange(2 * 8, dtype=torch.float32) * 0.001).reshape(2, 1, 8)
    r = x * (1 + scale) + shift
    assert abs(float(r[0, 0, 0])) < 1e-6 and abs(float(r[0, 0, 1]) - 0.102) < 1e-5


def test_attention_scale_known_answer():
    # tests/verify_rope_parity.rs:473-511: head_dim 64 -> scale 0.125
    assert float(np.float32(1.0) / np.sqrt(np.float32(64))) == 0.125


def test_calculate_shift_values():
    # SURVEY appendix / t2v_pipeline.rs:159-169
    for s, mu in ((384, 0.5217), (4992, 1.3017), (17556, 3.428)):
        assert abs(O.calculate_shift(s) - mu) < 1e-3
    assert O.calculate_shift(256) == pytest.approx(0.5, abs=1e-6) and O.calculate_shift(4096) == pytest.approx(1.15, abs=1e-6)


def test_distilled_sigma_schedule_closed_form():
    # configs.rs:232 sigmas, stretch-to-terminal 0.1: s' = 1 - (1-s)*(0.9/0.275); timesteps truncated (scheduler.rs:659)
    sig = [1.0, 0.9937, 0.9875, 0.9812, 0.9750, 0.9094, 0.7250]
    s = O.FlowMatchEulerScheduler()
    ts = s.set_timesteps(sigmas=sig, mu=0.0)
    want = [1 - (1 - x) * (0.9 / 0.275) for x in sig]
    assert np.abs(s.sigmas[:-1] - np.array(want)).max() < 1e-5 and s.sigmas[-1] == 0.0
    assert ts[0] == 1000 and ts == [int(np.float32(v) * 1000) if False else t for v, t in zip(want, ts)]
    assert all(isinstance(t, int) for t in ts) and ts[1] == 979


def test_scheduler_step_formula():
    # tests/verify_scheduler_parity.rs:405: prev = sample + (sigma_next - sigma) * v ; counter-based indexing
    s = O.FlowMatchEulerScheduler()
    ts = s.set_timesteps(sigmas=[1.0, 0.75, 0.5], mu=0.0)
    x = torch.randn(1, 6, 4); v = torch.randn(1, 6, 4)
    y = s.step(v, float(ts[0]), x)
    assert (y - (x + (float(s.sigmas[1]) - float(s.sigmas[0])) * v)).abs().max() < 1e-6
    y2 = s.step(v, float(ts[1]), y)        # second call uses the incremented counter, not a timestep lookup
    assert (y2 - (y + (float(s.sigmas[2]) - float(s.sigmas[1])) * v)).abs().max() < 1e-6


def test_pack_unpack_roundtrip_and_layout():
    # tests/verify_pipeline_parity.rs:93-113 (MSE < 1e-10); packed == channels-last for patch size 1
    x = torch.randn(2, 8, 3, 4, 5)
    p = O.pack_latents(x)
    assert torch.equal(p, x.permute(0, 2, 3, 4, 1).reshape(2, 60, 8))
    assert torch.equal(O.unpack_latents(p, 3, 4, 5), x)


def test_video_coords_closed_form():
    # t2v_pipeline.rs:798-847: f' = clamp(8f-7,0)/25 ; h' = 32h ; w' = 32w ; (f,h,w) order, f slowest
    vc = O.build_video_coords(1, 3, 2, 2)[0]
    assert vc.shape == (12, 3)
    assert torch.allclose(vc[:, 0], torch.tensor([0.0] * 4 + [1 / 25] * 4 + [9 / 25] * 4), atol=1e-7)
    assert vc[3].tolist() == [0.0, 32.0, 32.0] and vc[5].tolist()[1:] == [0.0, 32.0]


def test_upsampler_axis_order_kat(golden):
    # modelled on tests/vae_tests.rs:119-180: zero conv weight -> out = bias + tiled d2s residual, first frame dropped
    g = golden("oracle_ops.safetensors")
    x, y, bias = g["up_x"], g["up_y"], g["up_bias"]
    cin, cf = x.shape[1], y.shape[1]
    for (co, to, ho, wo) in [(0, 0, 0, 0), (3, 1, 2, 5), (2, 2, 3, 1), (1, 0, 1, 4)]:
        t, st = (to + 1) // 2, (to + 1) % 2
        h, sh, w, sw = ho // 2, ho % 2, wo // 2, wo % 2
        s = st * 4 + sh * 2 + sw
        want = bias[co * 8 + s] + x[0, (co % (cin // 8)) * 8 + s, t, h, w]
        assert abs(float(y[0, co, to, ho, wo]) - float(want)) < 1e-4


def test_rmsnorm_zero_input_and_eps():
    # RmsNorm f32 stats (ltx_transformer.rs:99-119): zero row stays zero, eps guards the division
    assert torch.equal(O.rms_norm(torch.zeros(2, 8), None, 1e-6), torch.zeros(2, 8))


# ---------- (c) independent torch implementations ----------
@pytest.mark.parametrize("causal", [False, True])
def test_conv3d_vs_torch_conv3d(causal):
    x = torch.randn(2, 8, 5, 6, 7); w = torch.randn(12, 8, 3, 3, 3) / 10; b = torch.randn(12)
    xp = F.pad(x, (0, 0, 0, 0, 2, 0) if causal else (0, 0, 0, 0, 1, 1), mode="replicate")
    ref = F.conv3d(F.pad(xp, (1, 1, 1, 1, 0, 0)), w, b)
    assert (O.causal_conv3d(x, w, b, causal) - ref).abs().max() < 2e-4


def test_attention_vs_sdpa():
    D, H = 64, 4
    p = {f"a.{n}.weight": torch.randn(D, D) / 8 for n in ("to_q", "to_k", "to_v", "to_out.0")}
    p.update({f"a.{n}.bias": torch.randn(D) * 0.02 for n in ("to_q", "to_k", "to_v", "to_out.0")})
    p["a.norm_q.weight"] = torch.ones(D); p["a.norm_k.weight"] = torch.ones(D)
    x = torch.randn(2, 24, D); enc = torch.randn(2, 10, D)
    bias = torch.zeros(2, 1, 10); bias[:, :, 7:] = -10000.0
    y = O.attention(p, "a.", H, x, enc, bias, None)
    q = F.rms_norm(F.linear(x, p["a.to_q.weight"], p["a.to_q.bias"]), (D,), eps=1e-5).reshape(2, 24, H, 16).transpose(1, 2)
    k = F.rms_norm(F.linear(enc, p["a.to_k.weight"], p["a.to_k.bias"]), (D,), eps=1e-5).reshape(2, 10, H, 16).transpose(1, 2)
    v = F.linear(enc, p["a.to_v.weight"], p["a.to_v.bias"]).reshape(2, 10, H, 16).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias.unsqueeze(2)).transpose(1, 2).reshape(2, 24, D)
    ref = F.linear(o, p["a.to_out.0.weight"], p["a.to_out.0.bias"])
    assert rel_max(y, ref) < 1e-5


def test_norms_and_gelu_vs_torch():
    x = torch.randn(3, 5, 32)
    assert (O.rms_norm(x, None, 1e-6) - F.rms_norm(x, (32,), eps=1e-6)).abs().max() < 1e-5
    assert (O.layer_norm_no_params(x, 1e-6) - F.layer_norm(x, (32,), eps=1e-6)).abs().max() < 1e-5
    assert (O.gelu_approximate(x) - F.gelu(x, approximate="tanh")).abs().max() < 1e-6
    xc = torch.randn(1, 16, 2, 3, 3)
    assert (O.rms_norm_channels_first(xc) - F.rms_norm(xc.permute(0, 2, 3, 4, 1), (16,), eps=1e-8).permute(0, 4, 1, 2, 3)).abs().max() < 1e-5


def test_d2s_and_unpatchify_vs_einops():
    from einops import rearrange
    x = torch.randn(1, 16, 2, 3, 4)
    assert torch.equal(O.depth_to_space(x, 2, 2, 2), rearrange(x, "b (c p1 p2 p3) t h w -> b c (t p1) (h p2) (w p3)", p1=2, p2=2, p3=2))
    y = torch.randn(1, 48, 2, 3, 4)
    # vae.rs:1626-1654: H pairs with the FASTEST channel sub-index, W with the slower one
    assert torch.equal(O.unpatchify(y, 4, 1), rearrange(y, "b (c pt pw ph) f h w -> b c (f pt) (h ph) (w pw)", pt=1, pw=4, ph=4))


def test_rope_rotation_is_complex_multiply():
    cos, sin = O.rope_cos_sin(64, 1, 2, 3, 3, None, O.build_video_coords(1, 2, 3, 3))
    assert cos.shape == (1, 18, 64) and torch.all(cos[..., :4] == 1) and torch.all(sin[..., :4] == 0)   # 64 % 6 = 4 left-pad
    assert torch.equal(cos[..., 4::2], cos[..., 5::2])                                                    # repeat_interleave(2)
    x = torch.randn(1, 18, 64)
    y = O.apply_rotary_emb(x, cos, sin)
    z = torch.view_as_complex(x.reshape(1, 18, 32, 2)) * torch.complex(cos[..., ::2], sin[..., ::2])
    assert (y - torch.view_as_real(z).reshape(1, 18, 64)).abs().max() < 1e-5
    # freq layout: index fi = step*3 + axis (ltx_transformer.rs:495-498)
    vc = O.build_video_coords(1, 2, 3, 3)[0]
    ang = (vc[7, 1] / 2048 * 2 - 1) * (math.pi / 2) * 10000 ** (2 / 9)          # row 7, axis h, step 2 of 10
    assert abs(float(cos[0, 7, 4 + 2 * (2 * 3 + 1)]) - math.cos(float(ang))) < 2e-3


def test_pcg32_vectorised_equals_scalar(golden):
    g = golden("oracle_ops.safetensors")
    r = O.Pcg32(42, 1442695040888963407)
    assert [r.next_u32() for _ in range(16)] == g["pcg_u32"].tolist()
    a = O.Pcg32(7, 1442695040888963407).randn((3, 5, 7))
    r2 = O.Pcg32(7, 1442695040888963407)
    b = torch.tensor([v for _ in range(53) for v in r2.next_gaussian()])[:105].reshape(3, 5, 7)
    assert torch.equal(a, b) and abs(float(a.mean())) < 0.3 and 0.7 < float(a.std()) < 1.3


def test_tiled_decode_equals_untiled_when_tile_covers_everything():
    cfg = O.VaeConfig(latent_channels=8, decoder_block_out_channels=(32, 64, 128), decoder_layers_per_block=(1, 1, 1, 1))
    w = O.synth_weights(O.vae_decoder_weight_shapes(cfg), seed=3)
    z = torch.randn(1, 8, 2, 2, 2)
    a = O.vae_decode(w, cfg, z, torch.tensor([0.05]), use_tiling=True, use_framewise_decoding=True)
    b = O.decoder_forward(w, cfg, z, torch.tensor([0.05]))
    assert torch.equal(a, b)


# ---------- (d) drift guard against the committed oracle fixtures ----------
@pytest.mark.parametrize("name", ["A", "B", "C"])
def test_dit_fixture_reproduces(golden, name):
    import ast
    from safetensors import safe_open
    import os
    from conftest import GOLDEN
    g = golden(f"oracle_dit_{name}.safetensors")
    with safe_open(os.path.join(GOLDEN, f"oracle_dit_{name}.safetensors"), "pt") as f:
        md = f.metadata()
    cfg = O.DitConfig(**ast.literal_eval(md["cfg"]))
    Fr, H, W = ast.literal_eval(md["grid"])
    w = {k[2:]: v for k, v in g.items() if k.startswith("w.")}
    y = O.dit_forward(w, cfg, g["hidden"], g["enc"], g["timestep"], g.get("mask"), Fr, H, W, ast.literal_eval(md["rope_scale"]),
                      g.get("coords"), g.get("skip_layer_mask"), ast.literal_eval(md["skip_blocks"]))
    assert rel_max(y, g["out_f32"]) < 1e-4       # same code, same machine class: only thread-count reduction-order noise
    # NB the bf16 path rounds the TIMESTEP to bf16 first (ltx_transformer.rs:1051: 918 -> 920, 979 -> 980), so
    # out_bf16 is only comparable with an f32 run at the rounded timestep; that comparison lives in the GPU suite.
    assert torch.isfinite(g["out_bf16"]).all()


def test_vae_fixture_reproduces(golden):
    import ast
    g = golden("oracle_vae.safetensors")
    from tools_cfg import VAE_CFG
    cfg = O.VaeConfig(**VAE_CFG)
    w = O.synth_weights(O.vae_decoder_weight_shapes(cfg), seed=7)
    ck = torch.tensor([sum(float(v.double().sum()) for v in w.values()), sum(float(v.double().abs().sum()) for v in w.values())], dtype=torch.float64)
    assert torch.allclose(ck, g["weights_checksum"], rtol=1e-9), "torch RNG stream differs from the one the fixture was made with"
    y = O.decoder_forward(w, cfg, g["z"], g["timestep"])
    assert rel_max(y, g["out_f32"]) < 1e-4


def test_stochastic_sampling_step_formula():
    """scheduler.rs:557-575: x0 = x - sigma v; x' = (1 - sigma') x0 + sigma' noise, with the draw supplied."""
    sched = O.FlowMatchEulerScheduler(O.SchedulerCfg(stochastic_sampling=True))
    ts = sched.set_timesteps(sigmas=[1.0, 0.75, 0.5], mu=0.0)
    g = torch.Generator().manual_seed(0)
    x, v, nz = torch.randn(2, 6, 4, generator=g), torch.randn(2, 6, 4, generator=g), torch.randn(2, 6, 4, generator=g)
    s0, s1 = float(sched.sigmas[0]), float(sched.sigmas[1])
    out = sched.step(v, float(ts[0]), x, nz)
    assert torch.allclose(out, (1.0 - s1) * (x - s0 * v) + s1 * nz, atol=1e-6)
    # last step: sigma' = 0 -> the result is the clean-sample estimate, noise drops out
    sched2 = O.FlowMatchEulerScheduler(O.SchedulerCfg(stochastic_sampling=True))
    sched2.set_timesteps(sigmas=[0.5], mu=0.0)
    out2 = sched2.step(v, float(sched2.timesteps[0]), x, nz)
    assert torch.allclose(out2, x - float(sched2.sigmas[0]) * v, atol=1e-6)


def test_oracle_pinned_by_three_more_reference_scripts(golden):
    """scripts/test_unpatchify.py, verify_rng.py, test_rope_rotation.py executed unmodified (tools/gen_fixtures.py ref)."""
    u = golden("ref_unpatchify.safetensors")
    assert torch.equal(O.unpatchify(u["x"], 4, 1), u["out"])                         # vae.rs:1626-1654 axis order, exact
    r = golden("ref_rope_rotation.safetensors")
    got = O.apply_rotary_emb(r["x"], r["cos"], r["sin"])
    assert (got - r["out"]).abs().max() < 1e-6 and (got - r["out_diffusers"]).abs().max() < 1e-6


# ---------- (a, continued) the RoPE table and grid against the reference's own rope scripts ----------
ROPE_BASE = (20.0, 2048.0, 2048.0)                    # LtxVideoRotaryPosEmbed base sizes (ltx_transformer.rs:341-371)


def _mse(a, b):
    return float(((a.double() - b.double()) ** 2).mean())


@pytest.mark.parametrize("dim", [2048, 4096])
def test_rope_table_matches_reference_scripts(golden, dim):
    """LtxVideoRotaryPosEmbed::forward (ltx_transformer.rs:436-524) against scripts/compare_rope_freqs.py
    (rust_compute_freqs AND diffusers_compute_freqs) and scripts/debug_rope.py, executed unmodified by
    tools/gen_fixtures.py rope.  Bar: tests/verify_rope_parity.rs:253-254, MSE < 1e-5 (the angles reach 1.5e4 rad in f32).
    Pins the frequency layout (per frequency: f, h, w), transpose-then-flatten, repeat_interleave 2 and the LEFT pad of
    dim % 6 columns (2 at dim 2048, 4 at 4096) with cos = 1, sin = 0."""
    r = golden("ref_rope_table.safetensors")
    F_, H_, W_ = (int(v) for v in r["fhw"])
    scale = tuple(float(v) for v in r["scale"])
    # (1) the grid path: prepare_video_coords (:373-433) with the pipeline's interpolation scale
    cos, sin = O.rope_cos_sin(dim, 1, F_, H_, W_, scale, None)
    wc, ws = r[f"rust_cos_{dim}_scaled"], r[f"rust_sin_{dim}_scaled"]
    assert cos.shape == wc.shape == (1, F_ * H_ * W_, dim)
    assert _mse(cos, wc) < 1e-5 and _mse(sin, ws) < 1e-5
    pad = dim % 6
    assert pad in (2, 4) and torch.equal(cos[..., :pad], wc[..., :pad]) and torch.equal(sin[..., :pad], ws[..., :pad])
    assert torch.equal(wc[..., :pad], torch.ones_like(wc[..., :pad])) and not torch.equal(wc[..., pad:pad + 2], torch.ones_like(wc[..., :2]))
    assert _mse(cos[..., 0::2], r[f"diffusers_cos_{dim}_scaled_even"]) < 1e-5 and torch.equal(cos[..., 0::2], cos[..., 1::2])
    assert _mse(sin[..., 0::2], r[f"diffusers_sin_{dim}_scaled_even"]) < 1e-5 and torch.equal(sin[..., 0::2], sin[..., 1::2])
    if dim == 2048:
        assert _mse(cos[..., 0::2], r["debug_cos_2048_scaled_even"]) < 1e-5 and _mse(sin[..., 0::2], r["debug_sin_2048_scaled_even"]) < 1e-5
    # small angles (the first frequencies) agree far tighter than the bar: not just "both look like noise"
    assert (cos[..., pad:pad + 96] - wc[..., pad:pad + 96]).abs().max() < 1e-5
    # (2) the video_coords path (:449-463): coords / base = the script's fractional grid
    grid = r["grid_rand"]
    coords = grid * torch.tensor(ROPE_BASE)
    cos2, sin2 = O.rope_cos_sin(dim, 1, 1, 1, grid.shape[1], None, coords)
    assert _mse(cos2, r[f"rust_cos_{dim}_rand"]) < 1e-5 and _mse(sin2, r[f"rust_sin_{dim}_rand"]) < 1e-5
    assert (cos2[..., pad:pad + 96] - r[f"rust_cos_{dim}_rand"][..., pad:pad + 96]).abs().max() < 1e-5


def test_rope_grid_order_matches_reference_scripts(golden):
    """scripts/compare_rope_grid.py (rust_expected_grid, diffusers_rope_grid) and scripts/debug_rope.py
    (prepare_video_coords_debug): the grid is [B, F*H*W, 3] with columns (f, h, w), f slowest.  The oracle builds the grid
    inside rope_cos_sin, so it is read back through a dim-6 table: one frequency pi/2, columns (f, f, h, h, w, w), and
    angle = pi/2 (2 g - 1) is invertible for g in [0, 1]."""
    r = golden("ref_rope_table.safetensors")
    F_, H_, W_ = (int(v) for v in r["fhw"])

    def grid_of(scale, batch=1):
        cos, sin = O.rope_cos_sin(6, batch, F_, H_, W_, scale, None)
        ang = torch.atan2(sin, cos)[..., 0::2]                       # [B, S, 3]
        return (ang / (math.pi / 2) + 1) / 2

    frac = grid_of(tuple(b / max(n - 1, 1) for b, n in zip(ROPE_BASE, (F_, H_, W_))))      # scale * 1 / base = 1 / (extent - 1)
    assert (frac - r["grid_frac"]).abs().max() < 1e-6
    raw = grid_of(tuple(b * 0.125 for b in ROPE_BASE))                                      # g = index / 8 stays inside [0, 1]
    assert (raw * 8 - r["grid_raw"]).abs().max() < 1e-5
    scaled = grid_of(tuple(float(v) for v in r["scale"]), batch=2)
    assert scaled.shape == r["grid_scaled"].shape and (scaled - r["grid_scaled"]).abs().max() < 1e-6


def test_pcg32_gaussians_match_reference_test_rng_script(golden):
    """scripts/test_rng.py (its own copy of Pcg32, both Box-Muller values kept, f64 math): first ten values for the seed
    and increment of main.rs:568; utils/deterministic_rng.rs:44-81 computes the same in f32."""
    want = golden("ref_rope_table.safetensors")["test_rng_values"]
    got = O.Pcg32(42, 1442695040888963407).randn((10,))
    assert (got.double() - want).abs().max() < 1e-5


# ---------- round 5: the two decoder variants (vae.rs:1212-1236, 676-689 / 741-753) ----------
def test_spatial_only_upsampler_vs_pixel_shuffle():
    """The (1, 2, 2) upsampler: per frame it is torch's pixel_shuffle(2) (channel c*4 + sh*2 + sw -> pixel (2h + sh, 2w + sw)),
    no frame dropped, residual = the same shuffle of x with its channels repeated 4 / upsample_factor times."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 16, 3, 4, 5, generator=g)
    p = {"u.conv.conv.weight": torch.randn(32, 16, 3, 3, 3, generator=g) / 20, "u.conv.conv.bias": torch.randn(32, generator=g)}
    y = O.upsampler(p, "u.", x, 8, False, stride=(1, 2, 2), residual=True)
    h = O.causal_conv3d(x, p["u.conv.conv.weight"], p["u.conv.conv.bias"], False)
    want = torch.stack([F.pixel_shuffle(h[:, :, t], 2) for t in range(3)], 2)
    res = torch.stack([F.pixel_shuffle(x[:, :, t], 2) for t in range(3)], 2).repeat(1, 2, 1, 1, 1)     # 16 / 4 = 4 channels -> 8
    assert y.shape == (2, 8, 3, 8, 10) and torch.allclose(y, want + res, atol=1e-6)
    assert torch.allclose(O.upsampler(p, "u.", x, 8, False, stride=(1, 2, 2), residual=False), want, atol=1e-6)


def test_noise_injection_formula_and_plane_stream():
    """maybe_inject_noise: x + plane[h, w] * scale[c], one plane per injection broadcast over batch, channels and frames; the planes
    are Pcg32(seed, k).randn((H, W)), k counting injections; no scale in the checkpoint or no stream = identity."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 3, 5, 6, generator=g); sc = torch.randn(4, 1, 1, generator=g)
    n = O.NoisePlanes(11)
    y0 = O._inject_noise(x, sc, n); y1 = O._inject_noise(x, sc, n)
    p0, p1 = O.Pcg32(11, 0).randn((5, 6)), O.Pcg32(11, 1).randn((5, 6))
    for y, pl in ((y0, p0), (y1, p1)):
        assert torch.allclose(y, x + pl[None, None, None] * sc.reshape(1, 4, 1, 1, 1), atol=1e-6)
    assert n.k == 2 and torch.equal(O._inject_noise(x, None, n), x) and torch.equal(O._inject_noise(x, sc, None), x) and n.k == 2
    # a decoder whose flags are set: the shapes function names the scales the reference looks up, and only flagged blocks draw
    cfg = O.VaeConfig(latent_channels=8, decoder_block_out_channels=(32, 64), decoder_layers_per_block=(1, 1, 1), decoder_upsample_factor=(2, 2),
                      decoder_upsample_residual=(True, True), decoder_inject_noise=(True, False, False), timestep_conditioning=False)
    w = O.synth_weights(O.vae_decoder_weight_shapes(cfg), seed=1)
    assert sorted(k for k in w if "per_channel" in k) == ["up_blocks.1.resnets.0.per_channel_scale1.weight", "up_blocks.1.resnets.0.per_channel_scale2.weight"]
    n = O.NoisePlanes(1)
    z = torch.randn(1, 8, 2, 3, 3, generator=g)
    out = O.decoder_forward(w, cfg, z, None, noise=n)
    assert n.k == 2 and not torch.allclose(out, O.decoder_forward(w, cfg, z, None))
