"""GPU suite: a small-M linear layer's K ranges left to the row norm that follows it (round 5; GemmArgs::defer_parts,
RowNormArgs::parts).

At C1's 384 tokens the DiT's ff2 (K = 8192) is a latency-bound weight stream: the shape rule cuts its K into four ranges, which
run as separate blocks and leave their f32 sums in a buffer; the next block's norm1 (or the final LayerNorm) adds them in part
order, applies gate * y + h (LtxVideoTransformerBlock::forward, ltx_transformer.rs:929-934), writes h and normalises the row it
has just finished.  Same K partition, same order, same expressions as the in-launch reduction + epilogue + stand-alone norm:
the bar is bit equality of both h and the normalised rows."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


@pytest.mark.parametrize("M,N,K,B,kind,gated", [(384, 2048, 8192, 1, 0, True), (384, 2048, 8192, 2, 1, True), (128, 1024, 10240, 1, 0, False),
                                                (301, 1032, 8192, 1, 0, True), (96, 512, 2048, 1, 0, True)])
def test_deferred_ranges_give_the_bits_of_the_in_launch_reduction(hip, M, N, K, B, kind, gated):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(DEV)
    b = torch.randn(N, generator=g).bfloat16().to(DEV); h0 = torch.randn(M, N, generator=g).bfloat16().to(DEV)
    rpb = M // B if M % B == 0 else M
    nb = M // rpb
    gate = torch.randn(nb, N, generator=g).to(DEV) if gated else None
    scale = (0.1 * torch.randn(nb, N, generator=g)).to(DEV); shift = (0.1 * torch.randn(nb, N, generator=g)).to(DEV)
    P = hip.lib.ltx_op_linear_split_factor(M, N, K)
    assert P == (4 if K >= 8192 else 1)
    h_ref = hip.ops.linear(x, w, b, epi=2 if gated else 3, resid=h0, gate=gate, rows_per_batch=rpb)
    y_ref = hip.ops.rownorm(h_ref, kind=kind, eps=1e-6, scale=scale, shift=shift, rows_per_batch=rpb)
    parts = hip.ops.linear_deferred(x, w)
    assert parts.shape[0] == P
    h, y = hip.ops.rownorm_deferred(parts, b, h0, gate, kind=kind, eps=1e-6, scale=scale, shift=shift, rows_per_batch=rpb)
    torch.cuda.synchronize()
    assert torch.equal(h.view(torch.int16), h_ref.view(torch.int16))
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
    want = h0.float() + (gate.repeat_interleave(rpb, 0) if gated else 1.0) * (x.float() @ w.float().t() + b.float())
    assert (h.float() - want).norm() / want.norm() < 4e-3


def test_dit_with_few_tokens_defers_ff2_and_returns_the_same_bits(hip):
    """A two-layer DiT at the real width (D = 2048: ff2 has K = 8192) over 96 tokens, batch of two: ltx_dit_forward with the
    deferred reduction (default) and with ff2_defer=0 - identical outputs; with a skip-layer mask the deferral is off by itself."""
    import ltx_oracle as O
    cfgd = dict(in_channels=32, out_channels=32, num_attention_heads=32, attention_head_dim=64, cross_attention_dim=2048, num_layers=2, caption_channels=64)
    cfg = O.DitConfig(**cfgd)
    w = O.synth_weights(O.dit_weight_shapes(cfg), seed=5)
    Fr, H, W, K = 2, 6, 8, 16
    S = Fr * H * W
    g = torch.Generator().manual_seed(6)
    hidden = torch.randn(2, S, 32, generator=g); enc = torch.randn(2, K, 64, generator=g)
    mask = torch.ones(2, K); mask[0, 9:] = 0
    coords = O.build_video_coords(2, Fr, H, W)
    t = torch.tensor([896.0, 640.0])
    model = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(**cfgd), {k: v.to(DEV) for k, v in w.items()}, torch.bfloat16)
    args = (hidden.to(DEV), enc.to(DEV), t, mask.to(DEV), Fr, H, W, None, coords.to(DEV))
    hip.prof_enable(True)
    out = model.forward(*args)
    hip.prof_enable(False)
    with hip.options(ff2_defer="0"):
        ref = model.forward(*args)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out, ref)
    wr = {k: v.bfloat16().float() for k, v in w.items()}
    want = O.dit_forward(wr, cfg, hidden.bfloat16().float(), enc.bfloat16().float(), t, mask, Fr, H, W, None, coords)
    e = float((out.float().cpu() - want).norm() / want.norm())
    assert e <= 2e-2, e
    # gemm_off=big (a documented A/B arm): gemm.hip's kernel serves the layer and knows nothing of deferred ranges - the
    # deferral must switch itself off (ltx_gemm_defer_ok), not leave the row norm reading an unwritten parts buffer
    with hip.options(gemm_off="big"):
        alt = model.forward(*args)
    e = float((alt.float().cpu() - want).norm() / want.norm())
    assert e <= 2e-2, e


def test_deferred_linear_is_refused_when_its_kernel_family_is_off(hip):
    x = torch.randn(96, 8192).bfloat16().to(DEV); w = torch.randn(512, 8192).bfloat16().to(DEV)
    with hip.options(gemm_off="big"):
        with pytest.raises(hip.LtxError):
            hip.ops.linear_deferred(x, w)
    with hip.options(gemm_off="ring"):
        with pytest.raises(hip.LtxError):
            hip.ops.linear_deferred(x, w)
