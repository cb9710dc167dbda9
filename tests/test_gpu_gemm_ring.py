"""GPU suite: the small-M deep-ring GEMM (csrc/gemm_ring.hip, plan family ring:*) against gemm_big's tiles.

gemm_ring keeps gemm_big's K partition (the shape-only split factor, K-ranges [part * nk / sf, (part + 1) * nk / sf), the
canonical sum of the parts) and its MFMA, so every ring tile must return the same bits as gemm_big for the same call - the
condition for being one more plan of the per-shape measurement.  Shapes: the C1 DiT's linear layers (384 tokens), the 128 text
rows, ragged M / N / K, split and unsplit."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

RING_TILES = ["ring:96x64", "ring:96x96", "ring:96x128", "ring:64x64", "ring:64x128", "ring:128x64", "ring:128x128", "ring:128x96", "ring:96x32", "ring:128x32", "ring:64x32"]


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


def _arms(hip, tile, fn, kind=0):
    """fn() on gemm_big's static tile (every other family off, plan cache bypassed), then on the forced ring tile (kind: 0 linear, 1 conv)."""
    with hip.options(gemm_off="ring+asm16+halo+p8", gemm_tune="0"):
        ref = fn(False)
    with hip.options(gemm_plan=tile):
        hip.prof_enable(True)
        got = fn(True)
        ms, _, cnt = hip.prof_report_kernel(kind, hip.PROF_KERNELS.index("gemm_ring_kernel"))
        hip.prof_enable(False)
    torch.cuda.synchronize()
    assert cnt >= 1, "the forced ring tile did not run"
    return ref, got


@pytest.mark.parametrize("tile", RING_TILES)
@pytest.mark.parametrize("M,N,K,epi", [(384, 6144, 2048, 0), (384, 2048, 2048, 2), (384, 8192, 2048, 1), (384, 2048, 8192, 2),
                                       (128, 4096, 4096, 0), (128, 10240, 4096, 1), (128, 4096, 10240, 3),
                                       (301, 1028, 200, 0), (77, 36, 72, 3), (1, 2048, 256, 0), (1152, 2048, 2048, 0)])
def test_ring_tiles_bit_identical_to_gemm_big(hip, tile, M, N, K, epi):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    resid = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi in (2, 3) else None
    gate = torch.randn(1, N, device="cuda", generator=g) if epi == 2 else None
    ref, got = _arms(hip, tile, lambda forced: hip.ops.linear(x, w, b, epi=epi, resid=resid, gate=gate, rows_per_batch=M))
    assert torch.isfinite(got.float()).all()
    if epi == 0:                                                    # and it is a GEMM, not two equal wrongs
        ref32 = x.float() @ w.float().t() + b.float()
        assert (got.float() - ref32).norm() / ref32.norm() < 3e-3
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("tile", ["ring:96x96", "ring:64x128"])
def test_ring_segmented_qkv_output_and_two_batch_gate(hip, tile):
    """The fused q|k|v projection written as three dense matrices, and gate * y + residual with one gate row per batch element
    (two batch elements of 192 rows)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    M, K = 384, 2048
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(6144, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = torch.randn(6144, device="cuda", generator=g).bfloat16()
    ref, got = _arms(hip, tile, lambda forced: hip.ops.linear_segmented(x, w, b, 2048))
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    w2 = w[:2048].contiguous(); b2 = b[:2048].contiguous()
    resid = torch.randn(M, 2048, device="cuda", generator=g).bfloat16(); gate = torch.randn(2, 2048, device="cuda", generator=g)
    ref, got = _arms(hip, tile, lambda forced: hip.ops.linear(x, w2, b2, epi=2, resid=resid, gate=gate, rows_per_batch=192))
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    want = resid.float() + gate.repeat_interleave(192, 0) * (x.float() @ w2.float().t() + b2.float())
    assert (got.float() - want).norm() / want.norm() < 4e-3


def test_ring_plans_are_measured_and_saved_for_small_m(hip, tmp_path):
    """A fresh shape with M = 384 goes through the plan measurement with the ring family among the candidates; whatever wins, the
    result equals the static gemm_big tile's, and a saved plan file loads back."""
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 384, 2304, 2048
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    got = hip.ops.linear(x, w, b, epi=0)
    plan = hip.ops.gemm_plan(M, N, K)
    assert plan != ""
    with hip.options(gemm_off="ring", gemm_tune="0"):
        ref = hip.ops.linear(x, w, b, epi=0)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    p = str(tmp_path / "plans.txt")
    hip.plan_save(p)
    assert any(line.split()[:3] == [str(M), str(N), str(K)] for line in open(p) if not line.startswith("#"))
    hip.plan_load(p)


@pytest.mark.parametrize("tile", RING_TILES)
@pytest.mark.parametrize("M,N,K,epi", [(384, 2048, 2048, 2), (384, 8192, 2048, 1), (128, 4096, 10240, 3), (301, 1028, 200, 0), (77, 36, 72, 3), (5, 2048, 256, 0)])
def test_ring_on_packed_weights_bit_identical(hip, tile, M, N, K, epi):
    """The tile-contiguous weight copy (GemmArgs::Wp: [N/32][K/64][32][64], zero padded - what the models hand the small-M kernel)
    against the row-major weights: same values through another address map, every bit equal; ragged N (not a multiple of 32) and K
    (not a multiple of 64) included."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 1)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    resid = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi in (2, 3) else None
    gate = torch.randn(1, N, device="cuda", generator=g) if epi == 2 else None
    wp = hip.ops.ring_pack(w)
    # the packed image holds exactly the weights, in 32 x 64 blocks
    blocks = wp.view(-(-N // 32), -(-K // 64), 32, 64)
    assert torch.equal(blocks[0, 0, : min(32, N), : min(64, K)], w[: min(32, N), : min(64, K)])
    with hip.options(gemm_plan=tile):
        ref = hip.ops.linear(x, w, b, epi=epi, resid=resid, gate=gate, rows_per_batch=M)
        hip.prof_enable(True)
        got = hip.ops.linear_packed(x, w, wp, b, epi=epi, resid=resid, gate=gate, rows_per_batch=M)
        _, _, cnt = hip.prof_report_kernel(0, hip.PROF_KERNELS.index("gemm_ring_kernel"))
        hip.prof_enable(False)
    assert cnt >= 1
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


def test_ring_fuzz_random_shapes_tiles_and_epilogues(hip):
    """120 random problems (M 1..512, N a multiple of 4 from 32, K a multiple of 8 from 8, every epilogue, a random ring tile each,
    with and without the packed weight copy): bit-identical to gemm_big's static tile and within the bf16 bar of f32 torch."""
    import random
    rnd = random.Random(20261003)
    g = torch.Generator(device="cuda").manual_seed(99)
    for it in range(120):
        M = rnd.choice([1, 2, 7, 31, 32, 33, 95, 96, 97, 128, 200, 255, 256, 257, 383, 384, 385, 500, 512])
        N = 4 * rnd.randint(8, 700)
        K = 8 * rnd.randint(1, 520)
        epi = rnd.randint(0, 3)
        tile = rnd.choice(RING_TILES)
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device="cuda", generator=g).bfloat16() if rnd.random() < 0.8 else None
        resid = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi in (2, 3) else None
        nb = rnd.choice([1, 2]) if M % 2 == 0 else 1
        gate = torch.randn(nb, N, device="cuda", generator=g) if epi == 2 else None
        packed = rnd.random() < 0.5
        wp = hip.ops.ring_pack(w) if packed else None
        def run(forced):
            if packed and forced:
                return hip.ops.linear_packed(x, w, wp, b, epi=epi, resid=resid, gate=gate, rows_per_batch=M // nb)
            return hip.ops.linear(x, w, b, epi=epi, resid=resid, gate=gate, rows_per_batch=M // nb)
        ref, got = _arms(hip, tile, run)
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (it, M, N, K, epi, tile, packed)
        if epi == 0:
            ref32 = x.float() @ w.float().t() + (b.float() if b is not None else 0)
            assert (got.float() - ref32).norm() / ref32.norm() < 4e-3, (it, M, N, K, tile)


def _cl(x): return x.permute(0, 2, 3, 4, 1).contiguous()


CONV_TILES = [t for t in RING_TILES if int(t.split("x")[1]) >= 64]


@pytest.mark.parametrize("tile", CONV_TILES)
@pytest.mark.parametrize("B,T,H,W,Cin,Cout,causal,what", [
    (1, 4, 8, 12, 1024, 1024, False, "the VAE mid block at C1's latent size (384 voxels): eight K ranges of 54 steps"),
    (2, 2, 6, 8, 512, 1024, True, "an edge tile of the tiled decode, batch of two, causal padding"),
    (1, 3, 5, 7, 192, 200, False, "ragged everywhere: 105 voxels, 200 output channels, three 64-channel slices"),
    (1, 1, 16, 24, 128, 128, True, "one frame (every frame tap is the replicated frame), conv_in-like"),
])
def test_ring_conv_mode_bit_identical_to_gemm_big(hip, tile, B, T, H, W, Cin, Cout, causal, what):
    """gemm_ring's conv mode (planes of a few hundred voxels: an activation row per voxel re-read per tap through the deep ring, the
    tap bookkeeping in the producer waves) against gemm_big's conv mode: same K order, K partition and part sum -> the same bits,
    with a bias and with a residual; and a convolution (f32 torch within the bf16 bar)."""
    g = torch.Generator().manual_seed(B + T + H + W + Cin)
    x = torch.randn(B, Cin, T, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).bfloat16(); b = torch.randn(Cout, generator=g).bfloat16()
    r = torch.randn(B, Cout, T, H, W, generator=g).bfloat16()
    xc, rc = _cl(x).cuda(), _cl(r).cuda()
    ref, got = _arms(hip, tile, lambda forced: (hip.ops.conv3d(xc, w.cuda(), b.cuda(), causal), hip.ops.conv3d(xc, w.cuda(), b.cuda(), causal, resid=rc)), kind=1)
    for a, c in zip(got, ref):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a.view(torch.int16), c.view(torch.int16)), what
    pad = (0, 0, 0, 0, 2, 0) if causal else (0, 0, 0, 0, 1, 1)
    xp = torch.nn.functional.pad(x.float(), pad, mode="replicate")
    want = torch.nn.functional.conv3d(xp, w.float(), b.float(), padding=(0, 1, 1))
    e = float((got[0].float().cpu().permute(0, 4, 1, 2, 3) - want).norm() / want.norm())
    assert e < 4e-3, (what, e)
