"""GPU suite, part 1: every fused HIP kernel, driven through the C ABI (include/ltxhip_ops.h), against
the CPU oracle's restatement of the reference op it replaces.

Tolerances (stated once):
  f32 mode  : max|hip - oracle| / max|oracle| <= 1e-3   (north_star bar; in practice ~1e-6..1e-5)
  bf16 mode : inputs are rounded to bf16 first and the oracle is evaluated IN F32 on those rounded
              inputs; the HIP result (bf16 storage, f32 accumulate) must be within rel-L2 <= 1.5e-2
              (two bf16 roundings: eps_bf16 = 2^-8 = 3.9e-3 per rounding).
"""
import math

import pytest
import torch

import ltx_oracle as O
from conftest import rel_l2, rel_max

pytestmark = pytest.mark.gpu
F32_TOL = 1e-3
BF16_TOL = 1.5e-2
DT = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


def rnd(dt, *shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dt)


def check(out, ref, dt):
    out = out.float().cpu()
    assert torch.isfinite(out).all()
    if dt == torch.float32:
        assert rel_max(out, ref) <= F32_TOL, rel_max(out, ref)
    else:
        assert rel_l2(out, ref) <= BF16_TOL, rel_l2(out, ref)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(384, 256, 128), (130, 96, 72), (1, 512, 256), (10, 32, 32), (4992, 256, 2048)])
def test_linear_bias(hip, dt, M, N, K):
    x, w, b = rnd(dt, M, K), rnd(dt, N, K, scale=K ** -0.5), rnd(dt, N, scale=0.1)
    y = hip.ops.linear(x.cuda(), w.cuda(), b.cuda())
    check(y, O.linear(x.float(), w.float(), b.float()), dt)


@pytest.mark.parametrize("dt", DT)
def test_linear_epilogues(hip, dt):
    M, N, K, S = 256, 192, 160, 128
    x, w, b = rnd(dt, M, K), rnd(dt, N, K, scale=K ** -0.5), rnd(dt, N, scale=0.1)
    r = rnd(dt, M, N, seed=3)
    gate = rnd(torch.float32, M // S, N, seed=4)
    lin = O.linear(x.float(), w.float(), b.float())
    check(hip.ops.linear(x.cuda(), w.cuda(), b.cuda(), epi=1), O.gelu_approximate(lin), dt)
    want = r.float() + gate.repeat_interleave(S, 0) * lin
    check(hip.ops.linear(x.cuda(), w.cuda(), b.cuda(), epi=2, resid=r.cuda(), gate=gate.cuda(), rows_per_batch=S), want, dt)
    check(hip.ops.linear(x.cuda(), w.cuda(), b.cuda(), epi=3, resid=r.cuda()), r.float() + lin, dt)
    check(hip.ops.linear(x.cuda(), w.cuda(), None), O.linear(x.float(), w.float(), None), dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rows,D", [(200, 2048), (77, 128), (33, 16), (5, 4096)])
def test_rownorm_rms_modulate(hip, dt, rows, D):
    x = rnd(dt, rows, D, scale=3.0)
    B = 1
    scale, shift = rnd(torch.float32, B, D, seed=1), rnd(torch.float32, B, D, seed=2)
    y = hip.ops.rownorm(x.cuda(), 0, 1e-6, None, scale.cuda(), shift.cuda(), rows, 0)
    ref = O.rms_norm(x.float(), None, 1e-6) * (1 + scale) + shift
    check(y, ref, dt)
    y2 = hip.ops.rownorm(x.cuda(), 0, 1e-8, None, scale.cuda(), shift.cuda(), rows, 1)          # VAE: eps 1e-8 + SiLU
    check(y2, torch.nn.functional.silu(O.rms_norm(x.float(), None, 1e-8) * (1 + scale) + shift), dt)
    y3 = hip.ops.rownorm(x.cuda(), 0, 1e-6)                                                     # plain RMS
    check(y3, O.rms_norm(x.float(), None, 1e-6), dt)


@pytest.mark.parametrize("dt", DT)
def test_rownorm_layernorm_batched_mod(hip, dt):
    B, S, D = 2, 40, 256
    x = rnd(dt, B * S, D, scale=2.0) + 0.5
    scale, shift = rnd(torch.float32, B, D, seed=1), rnd(torch.float32, B, D, seed=2)
    y = hip.ops.rownorm(x.cuda(), 1, 1e-6, None, scale.cuda(), shift.cuda(), S, 0)
    ref = O.layer_norm_no_params(x.float().reshape(B, S, D), 1e-6) * (1 + scale[:, None]) + shift[:, None]
    check(y, ref.reshape(B * S, D), dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("grid,D", [((2, 8, 8), 2048), ((4, 8, 12), 128), ((2, 4, 4), 64), ((1, 2, 3), 32), ((2, 4, 6), 4096)])
def test_rope_table_and_qknorm_rope(hip, dt, grid, D):
    Fr, H, W = grid
    S = Fr * H * W
    coords = O.build_video_coords(1, Fr, H, W)
    cos, sin = O.rope_cos_sin(D, 1, Fr, H, W, None, coords)
    c, s = hip.ops.rope_table(1, Fr, H, W, D, coords=coords[0].cuda())
    # bar: tests/verify_rope_parity.rs:253-254 MSE < 1e-5 (angles reach ~1.5e4 rad in f32: ulp-level freq noise is visible)
    assert ((c.cpu() - cos[0, :, ::2]) ** 2).mean() < 1e-5 and ((s.cpu() - sin[0, :, ::2]) ** 2).mean() < 1e-5
    assert torch.equal(c.cpu()[:, : (D % 6) // 2], torch.ones(S, (D % 6) // 2))
    x = rnd(dt, S, D, scale=2.0)
    w = (1 + 0.1 * rnd(torch.float32, D, seed=9)).to(dt)
    y = hip.ops.qknorm_rope(x.cuda(), w.cuda(), 1e-5, c, s)
    # reference semantics with the SAME tables (isolates the kernel from trig noise)
    ref = O.apply_rotary_emb(O.rms_norm(x.float()[None], w.float(), 1e-5), c.cpu().repeat_interleave(2, -1)[None], s.cpu().repeat_interleave(2, -1)[None])[0]
    check(y, ref, dt)
    check(hip.ops.qknorm_rope(x.cuda(), w.cuda(), 1e-5), O.rms_norm(x.float(), w.float(), 1e-5), dt)


def test_rope_table_grid_path(hip):
    # prepare_video_coords path (ltx_transformer.rs:373-433) with rope_interpolation_scale (1,1,1), tests/verify_dit_parity.rs
    Fr, H, W, D = 3, 4, 5, 64
    cos, sin = O.rope_cos_sin(D, 2, Fr, H, W, (1.0, 1.0, 1.0), None)
    c, s = hip.ops.rope_table(2, Fr, H, W, D, rope_scale=(1.0, 1.0, 1.0))
    assert (c.cpu() - cos.reshape(-1, D)[:, ::2]).abs().max() < 1e-4 and (s.cpu() - sin.reshape(-1, D)[:, ::2]).abs().max() < 1e-4


def ref_attention(q, k, v, heads, scale, bias):
    B, Sq, D = q.shape
    hd = D // heads
    qf = q.float().reshape(B, Sq, heads, hd).transpose(1, 2)
    kf = k.float().reshape(B, -1, heads, hd).transpose(1, 2)
    vf = v.float().reshape(B, -1, heads, hd).transpose(1, 2)
    att = qf @ kf.transpose(-1, -2) * scale
    if bias is not None:
        att = att + bias[:, None, None, :]
    return (torch.softmax(att, -1) @ vf).transpose(1, 2).reshape(B, Sq, D)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("hd,heads,Sq,Sk,B,biased", [(64, 2, 384, 384, 1, False), (64, 3, 200, 128, 2, True), (16, 4, 256, 10, 1, False),
                                                     (32, 2, 130, 70, 1, True), (128, 2, 160, 192, 1, False), (16, 2, 2048, 2048, 1, False),
                                                     (64, 8, 300, 300, 2, False), (16, 16, 130, 200, 1, True)])
def test_attention(hip, dt, hd, heads, Sq, Sk, B, biased):
    D = hd * heads
    q, k, v = rnd(dt, B, Sq, D), rnd(dt, B, Sk, D, seed=1), rnd(dt, B, Sk, D, seed=2)
    bias = None
    if biased:
        bias = torch.zeros(B, Sk); bias[:, Sk // 4:] = -10000.0; bias[0, 3] = -2.5
    scale = 1.0 / math.sqrt(hd)
    o = hip.ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, scale, bias.cuda() if biased else None)
    check(o, ref_attention(q, k, v, heads, scale, bias), dt)


def test_attention_online_softmax_rescale_branch(hip):
    # force the running max to jump late (rule: a rare data-dependent path needs its own input)
    hd, heads, S = 64, 1, 320
    q = torch.randn(1, S, hd).bfloat16(); k = torch.randn(1, S, hd).bfloat16(); v = torch.randn(1, S, hd).bfloat16()
    k[0, 300] = q[0, 5] * 8.0            # key 300 (last tile) dominates query 5
    o = hip.ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, 0.125)
    ref = ref_attention(q, k, v, heads, 0.125, None)
    assert rel_l2(o.float().cpu(), ref) <= BF16_TOL
    assert (o.float().cpu()[0, 5] - v.float()[0, 300]).abs().max() < 0.05


def cl(x):            # NCTHW -> channels-last
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ncthw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("causal", [False, True])
def test_conv3d_known_answer(hip, golden, dt, causal):
    g = golden("oracle_ops.safetensors")
    x, w, b = g["conv_x"].to(dt), g["conv_w"].to(dt), g["conv_b"].to(dt)
    y = hip.ops.conv3d(cl(x).cuda(), w.cuda(), b.cuda(), causal)
    ref = O.causal_conv3d(x.float(), w.float(), b.float(), causal)
    check(ncthw(y), ref, dt)
    if dt == torch.float32:
        assert rel_max(ncthw(y).cpu(), g["conv_y_causal" if causal else "conv_y_noncausal"]) <= F32_TOL


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,Cin,Cout,T,H,W", [(1, 128, 128, 3, 9, 11), (2, 16, 40, 2, 5, 6), (1, 256, 64, 1, 16, 16),
                                             # edge tiles of the tiled decode: a few dozen voxels x the mid block's width (split-K tiles since round 4)
                                             (1, 1024, 1024, 1, 4, 12), (1, 1024, 1024, 2, 4, 12), (1, 512, 512, 1, 8, 24)])
def test_conv3d_shapes_and_residual(hip, dt, B, Cin, Cout, T, H, W):
    x, w, b = rnd(dt, B, Cin, T, H, W), rnd(dt, Cout, Cin, 3, 3, 3, scale=(27 * Cin) ** -0.5), rnd(dt, Cout, scale=0.1)
    r = rnd(dt, B, Cout, T, H, W, seed=5)
    ref = O.causal_conv3d(x.float(), w.float(), b.float(), False)
    check(ncthw(hip.ops.conv3d(cl(x).cuda(), w.cuda(), b.cuda(), False)), ref, dt)
    check(ncthw(hip.ops.conv3d(cl(x).cuda(), w.cuda(), b.cuda(), False, resid=cl(r).cuda())), ref + r.float(), dt)


@pytest.mark.parametrize("dt", DT)
def test_upsampler_axis_order_and_values(hip, golden, dt):
    g = golden("oracle_ops.safetensors")
    x = g["up_x"].to(dt)
    w = torch.zeros(32, 16, 3, 3, 3, dtype=dt)
    y = hip.ops.upsample3d(cl(x).cuda(), w.cuda(), g["up_bias"].to(dt).cuda())
    if dt == torch.float32:
        assert (ncthw(y).cpu() - g["up_y"]).abs().max() < 1e-3          # exact index algebra (vae.rs:1106-1123, 1142-1161)
    # random weights: conv + d2s + first-frame drop + tiled residual
    x2, w2, b2 = rnd(dt, 1, 32, 2, 3, 4), rnd(dt, 128, 32, 3, 3, 3, scale=0.03), rnd(dt, 128, scale=0.1)
    p = {"conv.conv.weight": w2.float(), "conv.conv.bias": b2.float()}
    ref = O.upsampler(p, "", x2.float(), 16, False)
    check(ncthw(hip.ops.upsample3d(cl(x2).cuda(), w2.cuda(), b2.cuda())), ref, dt)
    ref_nores = O.upsampler(p, "", x2.float(), 16, False, residual=False)
    check(ncthw(hip.ops.upsample3d(cl(x2).cuda(), w2.cuda(), b2.cuda(), residual=False)), ref_nores, dt)


@pytest.mark.parametrize("dt", DT)
def test_conv_out_unpatchify_postprocess(hip, dt):
    x, w, b = rnd(dt, 1, 16, 2, 3, 5), rnd(dt, 48, 16, 3, 3, 3, scale=0.05), rnd(dt, 48, scale=0.1)
    ref = O.unpatchify(O.causal_conv3d(x.float(), w.float(), b.float(), False), 4, 1)
    y = hip.ops.conv_out_unpatchify(cl(x).cuda(), w.cuda(), b.cuda())
    check(y, ref, dt)
    yp = hip.ops.conv_out_unpatchify(cl(x).cuda(), w.cuda(), b.cuda(), postprocess=True)
    assert (yp.cpu() - O.postprocess_video(y.cpu())).abs().max() < 1e-3


def test_conv_out_on_the_halo_tile_vs_oracle_and_per_tap_tile(hip):
    """conv_out at the decoder's real channel count (128 -> 48, unpatchify epilogue, vae.rs:1626-1654) on a ragged plane
    (19 x 37: partial patches on both edges, three frames: first / middle / last frame taps): the 64-wide halo-staged tile
    (round 4 default) against the f32 oracle at the bf16 bar and BIT-IDENTICAL to the per-tap 192 x 64 tile it replaced
    (option gemm_off=halo_out; same K order), with and without the post-processing epilogue."""
    dt = torch.bfloat16
    x, w, b = rnd(dt, 1, 128, 3, 19, 37), rnd(dt, 48, 128, 3, 3, 3, scale=0.02), rnd(dt, 48, scale=0.1)
    ref = O.unpatchify(O.causal_conv3d(x.float(), w.float(), b.float(), False), 4, 1)
    outs = {}
    for arm in ("0", None):
        with hip.options(gemm_off="halo_out" if arm == "0" else None):
            outs[arm] = (hip.ops.conv_out_unpatchify(cl(x).cuda(), w.cuda(), b.cuda()), hip.ops.conv_out_unpatchify(cl(x).cuda(), w.cuda(), b.cuda(), postprocess=True))
    check(outs[None][0], ref, dt)
    assert torch.equal(outs[None][0], outs["0"][0]) and torch.equal(outs[None][1], outs["0"][1])
    assert (outs[None][1].cpu() - O.postprocess_video(outs[None][0].cpu())).abs().max() < 1e-3


def test_guidance_and_scheduler_step_reference_vectors(hip, golden):
    g = golden("ref_guidance.safetensors")           # produced by the reference's scripts/gen_guidance_ref.py
    t, u, p = g["noise_pred_text"].cuda(), g["noise_pred_uncond"].cuda(), g["noise_pred_perturb"].cuda()
    gs, ss = float(g["guidance_scale"]), float(g["stg_scale"])
    assert (hip.guidance_combine(t, u, None, gs).cpu() - g["combined_cfg"]).abs().max() < 1e-5       # tests/verify_cfg_parity.rs:82-88
    assert (hip.guidance_combine(t, u, p, gs, 0.0, ss).cpu() - g["combined_final"]).abs().max() < 1e-5  # tests/verify_guidance_parity.rs:58-67
    resc = hip.guidance_combine(t, u, p, gs, 0.7, ss).cpu()
    want = O.guidance_combine(g["noise_pred_text"], g["noise_pred_uncond"], g["noise_pred_perturb"], gs, 0.7, ss)
    assert (resc - want).abs().max() < 1e-3 and ((resc - want) ** 2).mean() < 1e-6                  # tests/verify_cfg_parity.rs:148-157
    # bf16 predictions, f32 latents; Euler step x + (sigma_next - sigma) v
    s = hip.FlowMatchEulerDiscreteScheduler()
    ts = s.set_timesteps([1.0, 0.9937, 0.9875], 0.0)
    lat = torch.randn(1, 256, 128)
    out = s.step(t.bfloat16(), ts[0], lat.cuda()).cpu()
    ref = lat + (s.sigmas[1] - s.sigmas[0]) * t.bfloat16().float().cpu()
    assert (out - ref).abs().max() < 1e-6


@pytest.mark.parametrize("dim", [2, 3, 4])
@pytest.mark.parametrize("ashape,bshape,extent", [((9, 64, 64), (9, 64, 64), 32), ((9, 96, 96), (9, 32, 96), 32), ((9, 96, 96), (9, 96, 32), 32),
                                                  ((16, 8, 8), (8, 8, 8), 8), ((3, 5, 7), (3, 5, 7), 50)])
def test_tile_blend(hip, dim, ashape, bshape, extent):
    # blend_h / blend_v / blend_t (vae.rs:1927-2006) vs the oracle's restatement; dims other than `dim` must agree
    a = torch.randn(1, 3, *ashape); b = torch.randn(1, 3, *bshape)
    for d in (2, 3, 4):
        if d != dim and a.shape[d] != b.shape[d]:
            pytest.skip("blend needs equal extents off-axis")
    ref = O._blend(a, b, extent, dim)
    out = hip.ops.blend(a.cuda(), b.cuda(), dim, extent).cpu()
    assert (out - ref).abs().max() < 1e-6


def test_errors_are_reported_not_crashed(hip):
    x = torch.randn(8, 30, device="cuda")           # K=30 is not a multiple of the 16-byte chunk
    with pytest.raises(hip.LtxError, match="multiple"):
        hip.ops.linear(x, torch.randn(16, 30, device="cuda"))
    with pytest.raises(hip.LtxError, match="head_dim"):
        hip.ops.attention(torch.randn(1, 8, 48, device="cuda"), torch.randn(1, 8, 48, device="cuda"), torch.randn(1, 8, 48, device="cuda"), 2, 1.0)


@pytest.mark.parametrize("tile", ["256x256", "192x256", "128x256", "256x128", "192x128", "128x128",
                                  "160x128", "192x64", "160x256w16", "192x256w16", "320x256w16", "256x256w16", "p8:256", "p8:128"])
def test_big_tile_gemm_all_tiles_and_epilogues(hip, tile, monkeypatch):
    """gemm_big.hip (LDS-DMA staged, 8 waves) and gemm_p8.hip (phase-interleaved): every tile shape, ragged M/N/K tails, every epilogue."""
    hip.set_option("gemm_plan", tile)
    dt = torch.bfloat16
    M, N, K, S = 1300, 328, 200, 650
    x, w, b = rnd(dt, M, K), rnd(dt, N, K, scale=K ** -0.5), rnd(dt, N, scale=0.1)
    r = rnd(dt, M, N, seed=3); gate = rnd(torch.float32, M // S, N, seed=4)
    lin = O.linear(x.float(), w.float(), b.float())
    check(hip.ops.linear(x.cuda(), w.cuda(), b.cuda()), lin, dt)
    check(hip.ops.linear(x.cuda(), w.cuda(), b.cuda(), epi=1), O.gelu_approximate(lin), dt)
    check(hip.ops.linear(x.cuda(), w.cuda(), b.cuda(), epi=2, resid=r.cuda(), gate=gate.cuda(), rows_per_batch=S), r.float() + gate.repeat_interleave(S, 0) * lin, dt)
    check(hip.ops.linear(x.cuda(), w.cuda(), b.cuda(), epi=3, resid=r.cuda()), r.float() + lin, dt)
    # conv modes with M >= 1024 voxels
    xc, wc, bc = rnd(dt, 1, 24, 3, 20, 19), rnd(dt, 40, 24, 3, 3, 3, scale=0.04), rnd(dt, 40, scale=0.1)
    check(ncthw(hip.ops.conv3d(cl(xc).cuda(), wc.cuda(), bc.cuda())), O.causal_conv3d(xc.float(), wc.float(), bc.float(), False), dt)
    xu, wu, bu = rnd(dt, 1, 32, 3, 20, 19), rnd(dt, 64, 32, 3, 3, 3, scale=0.04), rnd(dt, 64, scale=0.1)   # residual repeats = 64/32 = 2
    pu = {"conv.conv.weight": wu.float(), "conv.conv.bias": bu.float()}
    check(ncthw(hip.ops.upsample3d(cl(xu).cuda(), wu.cuda(), bu.cuda())), O.upsampler(pu, "", xu.float(), 8, False), dt)
    wo, bo = rnd(dt, 48, 24, 3, 3, 3, scale=0.04), rnd(dt, 48, scale=0.1)
    check(hip.ops.conv_out_unpatchify(cl(xc).cuda(), wo.cuda(), bo.cuda()), O.unpatchify(O.causal_conv3d(xc.float(), wo.float(), bo.float(), False), 4, 1), dt)


@pytest.mark.gpu
def test_gemm_plans_are_bit_identical_and_tuner_caches(hip, monkeypatch):
    """Every GEMM plan (gemm_big tiles, gemm_p8) accumulates K in the same order with the same MFMA, so the plan the
    dispatcher measures best may change speed only: outputs must be bit-identical across forced plans and the tuned path.
    (The tail split-K of gemm_big sums K-ranges separately -- an f32 re-association, covered by test_gemm_tail_split_k --
    so it is switched off here.)"""
    hip.set_option("gemm_splitk", "0")
    dt = torch.bfloat16
    M, N, K = 2048, 384, 320
    x, w, b = rnd(dt, M, K).cuda(), rnd(dt, N, K, scale=K ** -0.5).cuda(), rnd(dt, N, scale=0.1).cuda()
    xc, wc, bc = cl(rnd(dt, 1, 64, 3, 24, 20)).cuda(), rnd(dt, 128, 64, 3, 3, 3, scale=0.03).cuda(), rnd(dt, 128, scale=0.1).cuda()
    hip.set_option("gemm_tune", "0")
    base_lin, base_conv = hip.ops.linear(x, w, b, epi=1), hip.ops.conv3d(xc, wc, bc)
    for tile in ["256x256", "192x256", "128x256", "256x128", "192x128", "160x128", "128x128", "192x64",
                 "160x256w16", "192x256w16", "320x256w16", "256x256w16"]:
        hip.set_option("gemm_plan", tile)
        assert torch.equal(hip.ops.linear(x, w, b, epi=1), base_lin), tile
        assert torch.equal(hip.ops.conv3d(xc, wc, bc), base_conv), tile
    for p8 in ["p8:256", "p8:128"]:
        hip.set_option("gemm_plan", p8)
        assert torch.equal(hip.ops.linear(x, w, b, epi=1), base_lin), p8
        assert torch.equal(hip.ops.conv3d(xc, wc, bc), base_conv), p8
    hip.set_option("gemm_plan", None)
    hip.set_option("gemm_tune", "1")
    assert hip.ops.gemm_plan(M, N, K) == ""
    assert torch.equal(hip.ops.linear(x, w, b, epi=1), base_lin)
    plan = hip.ops.gemm_plan(M, N, K)
    assert plan != ""
    assert torch.equal(hip.ops.linear(x, w, b, epi=1), base_lin) and hip.ops.gemm_plan(M, N, K) == plan
    assert torch.equal(hip.ops.conv3d(xc, wc, bc), base_conv)
    assert hip.ops.gemm_plan(3 * 24 * 20, 128, 64, 1, 27, 3, 24, 20) != ""


@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("Sq,Sk", [(384, 384), (200, 333), (4992, 704)])
def test_attention_prescaled_q(hip, Sq, Sk, hd):
    """DiT self-attention fast path: q' = bf16(q * scale*log2e) produced by the q-norm kernel, softmax in base 2 with the
    running max fed to the S^T MFMA chain as its initial accumulator.  Reference: f32 softmax of ln2 * (q' k^T)."""
    heads = 2
    sc = hd ** -0.5
    q, k, v = rnd(torch.bfloat16, 1, Sq, heads * hd), rnd(torch.bfloat16, 1, Sk, heads * hd, seed=1), rnd(torch.bfloat16, 1, Sk, heads * hd, seed=2)
    k[0, Sk - 3] = q[0, 5] * 6.0                        # late dominant key: forces the rescale branch on the last tile
    qp = (q.float() * (sc * 1.4426950408889634)).bfloat16()
    o = hip.ops.attention_prescaled(qp.cuda(), k.cuda(), v.cuda(), heads)
    ref = ref_attention(qp, k, v, heads, math.log(2.0), None)
    assert rel_l2(o.float().cpu(), ref) <= BF16_TOL, rel_l2(o.float().cpu(), ref)
    # and it is the same function as the generic kernel up to the single extra rounding of q
    o2 = hip.ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, sc)
    assert rel_l2(o.float().cpu(), o2.float().cpu()) <= 2 * BF16_TOL


def test_attention_prescaled_dma_ring_screen(hip):
    """The fast path stages K/V with LDS-DMA into a 3-tile ring synchronised by counted vmcnt + raw barriers: screen it
    over ragged/odd sizes and repeated launches against the register-staged generic kernel (rare-race detector)."""
    g = torch.Generator().manual_seed(77)
    heads, hd = 8, 64
    for it in range(24):
        Sq = int(torch.randint(1, 700, (1,), generator=g)); Sk = int(torch.randint(1, 1700, (1,), generator=g))
        q, k, v = [torch.randn(1, s, heads * hd, generator=g).bfloat16().cuda() for s in (Sq, Sk, Sk)]
        qp = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
        ref = hip.ops.attention(q, k, v, heads, 0.125).float()
        outs = [hip.ops.attention_prescaled(qp, k, v, heads).float() for _ in range(3)]
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), (Sq, Sk)       # deterministic
        assert rel_l2(outs[0].cpu(), ref.cpu()) <= 2 * BF16_TOL, (Sq, Sk, rel_l2(outs[0].cpu(), ref.cpu()))


@pytest.mark.parametrize("tile", ["192x128", "160x128", "128x128", "256x256", "192x64", "160x256w16", "256x256w16"])
def test_gemm_tail_split_k(hip, tile, monkeypatch):
    """gemm_big cuts the tiles of a partly filled last round into K-ranges that meet in an in-launch reduction (f32 slabs,
    one release/acquire per tile).  Shapes chosen so that the split is active for every tile; repeated launches reuse the
    self-resetting arrival counters; option gemm_splitk=0 is the unsplit reference (differences = f32 summation order only)."""
    hip.set_option("gemm_plan", tile)
    dt = torch.bfloat16
    M, N, K, S = 1500, 56 if tile == "192x64" else 600, 2048, 750
    x, w, b = rnd(dt, M, K).cuda(), rnd(dt, N, K, scale=K ** -0.5).cuda(), rnd(dt, N, scale=0.1).cuda()
    r = rnd(dt, M, N, seed=3).cuda(); gate = rnd(torch.float32, M // S, N, seed=4).cuda()
    lin = O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())
    outs = [hip.ops.linear(x, w, b, epi=2, resid=r, gate=gate, rows_per_batch=S) for _ in range(4)]
    want = r.float().cpu() + gate.cpu().repeat_interleave(S, 0) * lin
    for o in outs:
        check(o, want, dt)
        assert torch.equal(o, outs[0])                      # the reduction order is fixed (slabs summed by part index)
    with hip.options(gemm_splitk="0"):
        base = hip.ops.linear(x, w, b, epi=2, resid=r, gate=gate, rows_per_batch=S)
    assert rel_l2(outs[0].float().cpu(), base.float().cpu()) <= 4e-3
    xc, wc, bc = cl(rnd(dt, 1, 256, 3, 20, 19)).cuda(), rnd(dt, 64 if tile != "192x64" else 48, 256, 3, 3, 3, scale=0.01).cuda(), rnd(dt, 64 if tile != "192x64" else 48, scale=0.1).cuda()
    if tile != "192x64" and tile != "256x256":              # N = 64 needs a 128-wide tile here
        y = hip.ops.conv3d(xc, wc, bc)
        check(ncthw(y), O.causal_conv3d(ncthw(xc).float().cpu(), wc.float().cpu(), bc.float().cpu(), False), dt)


def test_split_k_reduction_stress_full_size_mid_block(hip):
    """The VAE mid-block conv at C2 size (M = 4992, N = 1024, K = 27 x 1024) is the production user of the split-K
    reduction: 150 launches must reproduce the first bit for bit (canonical sum order, slabs published/acquired across
    XCDs) and match a CPU f32 conv on a halo'd crop."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(1, 1024, 13, 16, 24, generator=g).bfloat16()
    w = (torch.randn(1024, 1024, 3, 3, 3, generator=g) / 166).bfloat16(); b = torch.randn(1024, generator=g).bfloat16()
    xc, wc, bc = cl(x).cuda(), w.cuda(), b.cuda()
    first = hip.ops.conv3d(xc, wc, bc)
    for _ in range(150):
        assert torch.equal(hip.ops.conv3d(xc, wc, bc), first)
    y = ncthw(first).float().cpu()
    crop = x[:, :, 4:9, 3:10, 5:14].float()
    ref = O.causal_conv3d(crop, w.float(), b.float(), False)
    assert rel_l2(y[:, :, 5:8, 4:9, 6:13], ref[:, :, 1:4, 1:6, 1:8]) <= 1.5e-2


def test_fuzz_linear_and_conv_shapes(hip):
    """Randomised shapes through the GEMM dispatcher (all tiles, split-K, buffer-addressed staging, ragged M/N/K, small and
    odd conv geometries) against CPU f32: 40 linear + 16 conv cases, bf16."""
    g = torch.Generator().manual_seed(2024)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for it in range(40):
        M, N, K = ri(1, 2600), 4 * ri(1, 160), 8 * ri(1, 140)
        epi = ri(0, 3)
        x = torch.randn(M, K, generator=g).bfloat16(); w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16(); b = (torch.randn(N, generator=g) * 0.1).bfloat16()
        r = torch.randn(M, N, generator=g).bfloat16(); gate = torch.randn(1, N, generator=g)
        lin = x.float() @ w.float().T + b.float()
        want = [lin, O.gelu_approximate(lin), r.float() + gate * lin, r.float() + lin][epi]
        got = hip.ops.linear(x.cuda(), w.cuda(), b.cuda(), epi=epi, resid=r.cuda() if epi >= 2 else None,
                             gate=gate.cuda() if epi == 2 else None, rows_per_batch=M)
        assert rel_l2(got.float().cpu(), want) <= BF16_TOL, (it, M, N, K, epi, rel_l2(got.float().cpu(), want))
    for it in range(16):
        Cin, Cout = 8 * ri(1, 24), 8 * ri(1, 24)
        T, H, W = ri(1, 5), ri(1, 33), ri(1, 33)
        causal = bool(ri(0, 1))
        x = torch.randn(1, Cin, T, H, W, generator=g).bfloat16(); w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5).bfloat16()
        b = (torch.randn(Cout, generator=g) * 0.1).bfloat16()
        got = ncthw(hip.ops.conv3d(cl(x).cuda(), w.cuda(), b.cuda(), causal=causal)).float().cpu()
        want = O.causal_conv3d(x.float(), w.float(), b.float(), causal)
        assert rel_l2(got, want) <= BF16_TOL, (it, Cin, Cout, T, H, W, causal, rel_l2(got, want))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("M,N,K,seg", [(4992, 6144, 2048, 2048), (300, 384, 128, 128), (77, 192, 64, 64)])
def test_linear_segmented_matches_plain_linear(dtype, M, N, K, seg):
    """The fused q|k|v projection written as dense per-segment matrices: bit-identical to slicing the plain output."""
    import torch
    from ltxhip import ops
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(M, K, device="cuda", generator=g).to(dt)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(dt)
    b = torch.randn(N, device="cuda", generator=g).to(dt)
    plain = ops.linear(x, w, b)
    segd = ops.linear_segmented(x, w, b, seg)
    assert segd.shape == (N // seg, M, seg)
    for j in range(N // seg):
        assert torch.equal(segd[j], plain[:, j * seg:(j + 1) * seg])
    with pytest.raises(Exception):
        ops.linear_segmented(x, w, b, 96)


@pytest.mark.gpu
@pytest.mark.parametrize("Sk", [1, 7, 31, 32, 33, 64, 100, 127, 128])
def test_attention_short_key_kernel_vs_cpu(hip, Sk, monkeypatch):
    """head_dim 64, <= 128 keys (the DiT's cross-attention): K/V resident in LDS, one-shot softmax.  Against the f32 CPU
    reference with and without the additive key mask, ragged query counts, batch 2; and against the generic tiled kernel."""
    heads, hd = 3, 64
    for Sq, B, biased in [(1, 1, False), (31, 2, True), (500, 2, True), (333, 1, False)]:
        q, k, v = rnd(torch.bfloat16, B, Sq, heads * hd, seed=Sk), rnd(torch.bfloat16, B, Sk, heads * hd, seed=Sk + 1), rnd(torch.bfloat16, B, Sk, heads * hd, seed=Sk + 2)
        bias = None
        if biased:
            bias = torch.zeros(B, Sk); bias[:, Sk // 3 + 1:] = -10000.0; bias[0, 0] = -1.5     # reference mask value (ltx_transformer.rs:1063)
        args = (q.cuda(), k.cuda(), v.cuda(), heads, 0.125, bias.cuda() if biased else None)
        o = hip.ops.attention(*args)
        ref = ref_attention(q, k, v, heads, 0.125, bias)
        assert rel_l2(o.float().cpu(), ref) <= BF16_TOL, (Sq, Sk, B, biased, rel_l2(o.float().cpu(), ref))
        with hip.options(attn_off="cross"):
            o_generic = hip.ops.attention(*args)
        assert rel_l2(o.float().cpu(), o_generic.float().cpu()) <= BF16_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("Sk", [129, 192, 193, 256, 257, 320, 321, 384, 449, 512, 513, 577, 640, 4992])
def test_attention_pipelined_kernel_tile_counts(hip, Sk, monkeypatch):
    """The two-tile pipelined head_dim-64 kernel (4-slot LDS-DMA ring, loop unrolled by four): every prologue / remainder
    path (3..10 tiles, full and ragged last tile) against the f32 CPU reference and the one-tile-at-a-time kernel."""
    heads, hd, Sq = 2, 64, 200
    q, k, v = rnd(torch.bfloat16, 1, Sq, heads * hd, seed=Sk), rnd(torch.bfloat16, 1, Sk, heads * hd, seed=Sk + 1), rnd(torch.bfloat16, 1, Sk, heads * hd, seed=Sk + 2)
    k[0, Sk - 2] = q[0, 7] * 6.0                        # late dominant key: rescale on the last tile, pending S^T shifted with it
    k[0, 70] = q[0, 9] * 5.0                            # and one on the second tile
    qp = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
    o = hip.ops.attention_prescaled(qp.cuda(), k.cuda(), v.cuda(), heads)
    ref = ref_attention(qp, k, v, heads, math.log(2.0), None)
    assert rel_l2(o.float().cpu(), ref) <= BF16_TOL, (Sk, rel_l2(o.float().cpu(), ref))
    with hip.options(attn_off="q64"):                # the pipelined kernel itself (attn_q64 serves the launch by default since round 2)
        o_pipe = hip.ops.attention_prescaled(qp.cuda(), k.cuda(), v.cuda(), heads)
    assert rel_l2(o_pipe.float().cpu(), ref) <= BF16_TOL, (Sk, rel_l2(o_pipe.float().cpu(), ref))
    with hip.options(attn_off="pipe+q64"):
        o1 = hip.ops.attention_prescaled(qp.cuda(), k.cuda(), v.cuda(), heads)
    assert rel_l2(o_pipe.float().cpu(), o1.float().cpu()) <= BF16_TOL and rel_l2(o.float().cpu(), o1.float().cpu()) <= BF16_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("bn", ["128", "256"])
@pytest.mark.parametrize("B,Cin,Cout,T,H,W,causal", [(1, 64, 256, 3, 20, 37, False), (2, 128, 256, 2, 16, 16, True), (1, 192, 512, 4, 33, 18, False),
                                                     (1, 128, 256, 1, 5, 7, True)])
def test_conv3d_halo_staged_kernel(hip, bn, B, Cin, Cout, T, H, W, causal, monkeypatch):
    """conv_halo.hip (activation patch + rim staged once per nine in-plane taps): bit-identical to the per-tap kernels
    (same K-step order and MFMA), within bf16 tolerance of the CPU oracle; ragged patches, batch, causal / non-causal
    temporal padding, residual and depth-to-space epilogues."""
    dt = torch.bfloat16
    x, w, b = rnd(dt, B, Cin, T, H, W), rnd(dt, Cout, Cin, 3, 3, 3, scale=(27 * Cin) ** -0.5), rnd(dt, Cout, scale=0.1)
    r = rnd(dt, B, Cout, T, H, W, seed=5)
    xc, rc = cl(x).cuda(), cl(r).cuda()
    hip.set_option("gemm_splitk", "0")
    with hip.options(gemm_off="halo"):
        base = hip.ops.conv3d(xc, w.cuda(), b.cuda(), causal)
        base_res = hip.ops.conv3d(xc, w.cuda(), b.cuda(), causal, resid=rc)
        base_up = hip.ops.upsample3d(xc, w.cuda(), b.cuda(), causal)
    hip.set_option("gemm_plan", "halo:" + bn)
    y = hip.ops.conv3d(xc, w.cuda(), b.cuda(), causal)
    assert torch.equal(y, base)
    assert torch.equal(hip.ops.conv3d(xc, w.cuda(), b.cuda(), causal, resid=rc), base_res)
    assert torch.equal(hip.ops.upsample3d(xc, w.cuda(), b.cuda(), causal), base_up)
    ref = O.causal_conv3d(x.float(), w.float(), b.float(), causal)
    check(ncthw(y), ref, dt)


@pytest.mark.gpu
def test_real_shape_plans_are_bit_identical(hip, monkeypatch):
    """At the DiT's and the VAE's own shapes (not the small test shapes): the 16-wave GEMM tiles, the wide epilogue and the
    halo-staged conv give exactly the bits of the 8-wave / fragment-store / per-tap forms."""
    dt = torch.bfloat16
    hip.set_option("gemm_splitk", "0")
    S = 4992
    x, w, b = rnd(dt, S, 2048).cuda(), rnd(dt, 2048, 2048, scale=2048 ** -0.5).cuda(), rnd(dt, 2048, scale=0.1).cuda()
    r, gate = rnd(dt, S, 2048, seed=3).cuda(), rnd(torch.float32, 1, 2048, seed=4).cuda()
    outs = []
    for tile, wide in [("160x128", "0"), ("160x256w16", "1"), ("256x256w16", "1"), ("320x256w16", "0"), ("192x128", "1")]:
        hip.set_option("gemm_plan", tile); hip.set_option("gemm_wide_epi", wide)
        outs.append((hip.ops.linear(x, w, b), hip.ops.linear(x, w, b, epi=2, resid=r, gate=gate, rows_per_batch=S), hip.ops.linear(x, w, b, epi=1)))
    for o in outs[1:]:
        assert all(torch.equal(a, c) for a, c in zip(o, outs[0]))
    hip.set_option("gemm_plan", None); hip.set_option("gemm_wide_epi", None)
    # the 128-channel VAE stage's plane (128 x 192) with 4 frames
    xc, wc, bc = cl(rnd(dt, 1, 128, 4, 128, 192)).cuda(), rnd(dt, 128, 128, 3, 3, 3, scale=(27 * 128) ** -0.5).cuda(), rnd(dt, 128, scale=0.1).cuda()
    rc = cl(rnd(dt, 1, 128, 4, 128, 192, seed=6)).cuda()
    with hip.options(gemm_off="halo"):
        base = hip.ops.conv3d(xc, wc, bc, True, resid=rc)
    hip.set_option("gemm_plan", "halo:128")
    assert torch.equal(hip.ops.conv3d(xc, wc, bc, True, resid=rc), base)
    hip.set_option("gemm_wide_epi", "0")
    assert torch.equal(hip.ops.conv3d(xc, wc, bc, True, resid=rc), base)


@pytest.mark.parametrize("vae_flavour,mult", [(False, 1.0), (True, 1000.0)])
def test_timestep_embedding_kernel_vs_oracle(hip, vae_flavour, mult):
    """§8 a6 / a23 standalone: the sinusoid kernel against get_timestep_embedding (ltx_transformer.rs:271-309) and the
    VAE's variant (vae.rs:172-198, timestep x timestep_scale_multiplier first).  f32: angles reach 1000 rad, where one
    f32 ulp of the angle is 6e-5, so the bar is 2e-4 absolute on values in [-1, 1] (the reference's own embedding test
    uses MSE < 1e-5); bf16: the timestep is rounded to bf16 first (:1051) and the result rounded once."""
    ts = [1000.0, 979.0, 500.0, 99.0, 0.05, 0.0] if not vae_flavour else [0.05, 0.025, 1.0, 0.0]
    t = torch.tensor(ts)
    want = O.vae_timestep_embedding(t * mult) if vae_flavour else O.get_timestep_embedding(t)
    got = hip.ops.timestep_embedding(ts, vae_flavour, mult).cpu()
    assert got.shape == (len(ts), 256)
    assert (got - want).abs().max() <= 2e-4, (got - want).abs().max()
    assert ((got - want) ** 2).mean() < 1e-9
    tb = t.bfloat16()
    want_b = (O.vae_timestep_embedding((tb * torch.tensor(mult).bfloat16()).float()) if vae_flavour else O.get_timestep_embedding(tb.float())).bfloat16()
    got_b = hip.ops.timestep_embedding(ts, vae_flavour, mult, dtype=torch.bfloat16).cpu()
    assert (got_b.float() - want_b.float()).abs().max() <= 2 ** -7       # one bf16 ulp at 1.0


def test_team_collectives_over_rccl_single_rank(hip):
    """include/ltxhip_team.h on the one GPU of this box: a team of one goes through the real librccl.so calls
    (ncclGetUniqueId, ncclCommInitRank, ncclAllGather, grouped ncclSend + ncclRecv to itself) - the all-gather of one rank
    and a self exchange are copies.  Multi-rank behaviour is the same code with nranks > 1 (one process per GPU)."""
    import ctypes as C
    ident = (C.c_char * 128)()
    assert hip.lib.ltx_team_unique_id(ident) == 0, hip.lib.ltx_last_error()
    assert any(b != b"\x00" for b in ident)
    t = C.c_void_p()
    assert hip.lib.ltx_team_create(ident, 1, 0, torch.cuda.current_device(), C.byref(t)) == 0, hip.lib.ltx_last_error()
    assert hip.lib.ltx_team_size(t) == 1 and hip.lib.ltx_team_rank(t) == 0
    x = torch.randn(1, 4992, 128, device="cuda")
    y = torch.zeros_like(x)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert hip.lib.ltx_team_allgather_f32(t, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(), s) == 0, hip.lib.ltx_last_error()
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    z = torch.zeros(3 * 512 * 768, device="cuda")
    strip = torch.randn(3 * 512 * 768, device="cuda")
    assert hip.lib.ltx_team_exchange_f32(t, C.c_void_p(strip.data_ptr()), strip.numel(), 0, C.c_void_p(z.data_ptr()), z.numel(), 0, s) == 0, hip.lib.ltx_last_error()
    torch.cuda.synchronize()
    assert torch.equal(z, strip)
    assert hip.lib.ltx_team_exchange_f32(t, None, 0, -1, None, 0, -1, s) == 0                 # no peers: nothing to do
    assert hip.lib.ltx_team_exchange_f32(t, C.c_void_p(strip.data_ptr()), 8, 1, None, 0, -1, s) == 1      # peer outside the team
    hip.lib.ltx_team_destroy(t)
