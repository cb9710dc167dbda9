"""GPU suite: the one-wave-per-SIMD GEMMs of csrc/gemm_asm.hip (plan family asm16:*; round 1's 32x32x16 kernel in experiment
builds) against the gemm_big tiles.

Both accumulate bf16 products in f32 in ascending k; the MFMA shapes differ (32x32x16 vs 16x16x32).  The matrix core's
accumulation is a k-ordered chain, so the two families are bit-identical - which is also why every gemm_big / gemm_p8
/ conv_halo plan of a shape gives the same bits (tests/test_gpu_determinism.py).  The bar here is exact equality."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


PLAN16 = {"asm256x256": "asm16:256x256", "asm160x256": "asm16:160x256", "asm320x256": "asm16:320x256"}
TILE_IDX = {"asm256x256": "0", "asm320x256": "1", "asm160x256": "2"}        # gemm_asm.hip kAsmTiles order (x_gemm_asm_tile)


@pytest.mark.parametrize("mode,tile", [("1", "asm256x256"), ("1", "asm320x256"), ("1", "asm160x256"), ("16", "asm256x256"), ("16", "asm160x256"), ("16", "asm320x256")])
@pytest.mark.parametrize("M,N,K,epi", [(4992, 2048, 2048, 0), (4992, 6144, 2048, 1), (3001, 4104, 192, 0), (4992, 2048, 8192, 3)])
def test_asm_tiles_bit_identical_to_gemm_big(hip, mode, tile, M, N, K, epi):
    """mode "16": the 16x16x32 loop (the shipped plan family asm16:*, forced with the gemm_plan option); "1": round 1's 32x32x16
    loop, compiled into experiment builds only (x_gemm_asm=1)."""
    if mode == "1" and not hip.has_experiments(): pytest.skip("the 32x32x16 kernel is compiled into experiment builds only (make experiments)")
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    resid = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi == 3 else None
    # the reference arm must be gemm_big: asm16 is the default plan family of these shapes since round 3, so it is taken out of
    # the choice (gemm_off=asm16) and the plan cache bypassed (gemm_tune=0: cached_or_tuned_plan returns before it looks at the
    # cache, which may hold an asm16 plan of this shape from an earlier test)
    with hip.options(gemm_off="asm16", gemm_tune="0"):
        ref = hip.ops.linear(x, w, b, epi=epi, resid=resid)
    if mode == "16":
        with hip.options(gemm_plan=PLAN16[tile]):
            got = hip.ops.linear(x, w, b, epi=epi, resid=resid)
    else:
        with hip.options(x_gemm_asm="1", x_gemm_asm_tile=TILE_IDX[tile]):
            got = hip.ops.linear(x, w, b, epi=epi, resid=resid)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    ref32 = (x.float() @ w.float().t() + b.float())
    if epi == 0: assert (got.float() - ref32).norm() / ref32.norm() < 3e-3      # and it is a GEMM, not two equal wrongs
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("tile", ["asm256x256", "asm160x256", "asm320x256"])
def test_asm16_gate_residual_and_segmented_output_bit_identical(hip, tile):
    """The remaining epilogues of the DiT on the 16x16x32 loop's wide (LDS-transposed) epilogue: gate * y + residual with one
    f32 gate row per batch element (ragged M: the last row tile is partial) and the q|k|v projection written as three dense
    matrices; bit-identical to gemm_big's wide epilogue and to its fragment-wise one (option gemm_wide_epi=0)."""
    g = torch.Generator(device="cuda").manual_seed(7)
    M, N, K = 2 * 1531, 2048, 2048
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    resid = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    gate = torch.randn(2, N, device="cuda", generator=g)
    w3 = (torch.randn(3 * N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16(); b3 = torch.randn(3 * N, device="cuda", generator=g).bfloat16()
    def run():
        return hip.ops.linear(x, w, b, epi=2, resid=resid, gate=gate, rows_per_batch=1531), hip.ops.linear_segmented(x, w3, b3, N)
    with hip.options(gemm_off="asm16", gemm_tune="0"):          # reference arms: gemm_big (asm16 out of the plans, static tile)
        ref = run()
        with hip.options(gemm_wide_epi="0"):
            narrow = run()
    with hip.options(gemm_plan=PLAN16[tile]):
        wide = run()
    torch.cuda.synchronize()
    want = resid.float() + gate.repeat_interleave(1531, 0) * (x.float() @ w.float().t() + b.float())
    assert (ref[0].float() - want).norm() / want.norm() < 4e-3
    for got in (wide, narrow):
        for a, r in zip(got, ref):
            assert torch.equal(a.view(torch.int16), r.view(torch.int16))


def _cl(x): return x.permute(0, 2, 3, 4, 1).contiguous()


@pytest.mark.parametrize("B,T,H,W,Cin,Cout,causal,what", [
    (1, 13, 16, 24, 1024, 1024, False, "the VAE mid block at C2's latent size: the shape rule cuts K into three ranges = one frame tap each"),
    (1, 5, 33, 48, 128, 1152, False, "unsplit, ragged in M (30.9 tiles) and N (4.5 tiles)"),
    (2, 3, 20, 20, 192, 1024, True, "batch of two, causal padding, K ranges that start inside a frame tap and inside a channel slice"),
    (1, 4, 9, 31, 64, 1024, False, "one 64-channel slice, planes narrower than a tile row"),
])
def test_asm16_conv_mode_bit_identical_to_gemm_big(hip, B, T, H, W, Cin, Cout, causal, what):
    """gemm_asm16_conv_kernel (the generated loop in conv mode: A rows re-staged per tap, validity masks and tap offsets rebuilt per
    K-step, the fetch position carried in scalars) against gemm_big's conv mode: same K order, same K partition, same canonical sum
    of the parts - the same bits, with a bias, with a residual, and (depth-to-space) on the upsampler; and within the bf16 bar of
    f32 torch."""
    g = torch.Generator().manual_seed(B + T + H + W + Cin)
    x = torch.randn(B, Cin, T, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).bfloat16(); b = torch.randn(Cout, generator=g).bfloat16()
    r = torch.randn(B, Cout, T, H, W, generator=g).bfloat16()
    xc, rc = _cl(x).cuda(), _cl(r).cuda()
    def run():
        return hip.ops.conv3d(xc, w.cuda(), b.cuda(), causal), hip.ops.conv3d(xc, w.cuda(), b.cuda(), causal, resid=rc)
    with hip.options(gemm_tune="0", gemm_off="asm16+halo+p8"):
        ref = run()
    hip.prof_enable(True)
    with hip.options(gemm_plan="asm16c:256x256"):
        got = run()
    _, _, cnt = hip.prof_report_kernel(1, hip.PROF_KERNELS.index("gemm_asm16_kernel"))
    hip.prof_enable(False)
    assert cnt == 2, "the forced conv-mode plan did not serve the launches"
    for a, c in zip(got, ref):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a.view(torch.int16), c.view(torch.int16)), what
    pad = (0, 0, 0, 0, 2, 0) if causal else (0, 0, 0, 0, 1, 1)
    xp = torch.nn.functional.pad(x.float(), pad, mode="replicate")
    want = torch.nn.functional.conv3d(xp, w.float(), b.float(), padding=(0, 1, 1))
    e = float((got[0].float().cpu().permute(0, 4, 1, 2, 3) - want).norm() / want.norm())
    assert e < 4e-3, (what, e)


def test_asm16_conv_mode_depth_to_space_epilogue(hip):
    """The upsampler's conv (vae.rs:1090-1169: 8 x channels, depth-to-space, residual) on the conv-mode loop vs gemm_big."""
    g = torch.Generator().manual_seed(11)
    B, T, H, W, Cin = 1, 3, 20, 19, 128
    x = torch.randn(B, Cin, T, H, W, generator=g).bfloat16()
    w = (torch.randn(8 * Cin, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).bfloat16(); b = torch.randn(8 * Cin, generator=g).bfloat16()
    xc = _cl(x).cuda()
    with hip.options(gemm_tune="0", gemm_off="asm16+halo+p8"):
        ref = hip.ops.upsample3d(xc, w.cuda(), b.cuda())
    with hip.options(gemm_plan="asm16c:256x256"):
        got = hip.ops.upsample3d(xc, w.cuda(), b.cuda())
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
