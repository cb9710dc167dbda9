"""GPU suite: the STREAM form of the head_dim-64 self-attention (round 6; csrc/attn_q64.hip attn_q64_stream_kernel, option
attn_q64_stream=1; NOT the default - measured 4 - 6 % behind the block grid, docs/lab_notes.md R6.3): one persistent workgroup per CU walks a host-built item list - key-range parts of the blocks that
make the last, partial round of the grid (merged in the launch by the last arriver), whole 256-query blocks, the 128-query block -
and every item's first four key tiles and Q^T are requested by the item before it.  Replaces candle-flash-attn at
ltx_transformer.rs:699-712 on the DiT's own shapes (32 heads, S = 4992: more blocks than CUs).

Bars: rel-L2 <= 5e-3 against an f64 softmax(q k^T) v on sampled heads (the block grid's bar, test_gpu_attn_q64.py); against the
block grid itself <= 3e-3 (whole blocks run the same arithmetic - the parts' merge rounds differently); repeated launches
bit-identical (the schedule is static, parts are merged in part order whoever arrives last); the exact-max pass behind the
overflow check still runs from inside the stream (and the items after it start cold)."""
import math

import pytest
import torch

from conftest import rel_l2
from test_gpu_attn_q64 import mk, ref_attn_heads, TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


def run(hip, qp, k, v, heads, **opts):
    opts.setdefault("attn_q64_stream", "1")
    with hip.options(**opts):
        o = hip.ops.attention_prescaled(qp, k, v, heads)
        torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    return o


@pytest.mark.parametrize("B,S,heads", [(1, 4992, 32), (2, 4992, 32), (3, 4992, 32), (1, 2048, 64), (1, 1664, 96), (1, 9984, 16)])
def test_stream_vs_reference_and_block_grid(hip, B, S, heads):
    qp, k, v = (t.cuda() for t in mk(B, S, S, heads, seed=S + heads + B))
    o = run(hip, qp, k, v, heads)
    o2 = run(hip, qp, k, v, heads)
    assert torch.equal(o, o2)
    og = run(hip, qp, k, v, heads, attn_q64_stream="0")
    d = rel_l2(o.float().cpu(), og.float().cpu())
    pick = sorted({0, 5, heads // 2 + 1, heads - 1})
    errs = []
    for b in range(B):
        ref = ref_attn_heads(qp[b:b + 1], k[b:b + 1], v[b:b + 1], heads, pick)
        for hd_i in pick:
            errs.append(rel_l2(o[b:b + 1, :, hd_i * 64:hd_i * 64 + 64].float().cpu(), ref[hd_i].cpu()))
    print({"B": B, "S": S, "heads": heads, "stream_vs_f64_max": round(max(errs), 5), "stream_vs_grid": round(d, 6)})
    assert max(errs) <= TOL, max(errs)
    assert d <= 3e-3, d


def test_stream_on_column_slices_of_a_fused_buffer(hip):
    """q | k | v as column slices of one [S, 3 D] matrix (the dense_qkv=0 layout): other strides, other item offsets"""
    heads, S = 32, 4992
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(1, S, 3 * heads * 64, generator=g)
    qkv[..., :heads * 64] *= 0.125 * 1.4426950408889634
    qkv = qkv.bfloat16().cuda()
    D = heads * 64
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    o = run(hip, q, k, v, heads)
    og = run(hip, q, k, v, heads, attn_q64_stream="0")
    assert rel_l2(o.float().cpu(), og.float().cpu()) <= 3e-3
    ref = ref_attn_heads(q.contiguous(), k.contiguous(), v.contiguous(), heads, [3, 30])
    for hd_i in (3, 30):
        assert rel_l2(o[..., hd_i * 64:hd_i * 64 + 64].float().cpu(), ref[hd_i].cpu()) <= TOL


def test_stream_overflow_takes_the_exact_pass_and_goes_on(hip):
    """A late key far above the first tile's maximum in a few (head, query) pairs: the items that hold them re-run with exact row maxima
    (counter), the items behind them start cold, every row still matches the reference."""
    heads, S = 32, 4992
    qp, k, v = mk(1, S, S, heads, seed=77, qscale=2.0)
    qf, kf = qp.float(), k.float()
    for (qrow, krow) in ((17, 4000), (2500, 4900), (4990, 3000)):
        kf[0, krow] = qf[0, qrow] * 30.0
    k = kf.bfloat16()
    qp, k, v = qp.cuda(), k.cuda(), v.cuda()
    hip.attention_fallback_counts(reset=True)
    o = run(hip, qp, k, v, heads)
    fb = hip.attention_fallback_counts(reset=True)
    assert fb[0] > 0, fb
    o2 = run(hip, qp, k, v, heads)
    assert torch.equal(o, o2)
    ref = ref_attn_heads(qp, k, v, heads, [0, 13, 31])
    for hd_i in (0, 13, 31):
        got = o[..., hd_i * 64:hd_i * 64 + 64].float().cpu()
        assert rel_l2(got, ref[hd_i].cpu()) <= TOL
        for qrow in (17, 2500, 4990):
            assert rel_l2(got[0, qrow], ref[hd_i][0, qrow].cpu()) <= 2e-2
    print({"fallback_workgroups": fb[0]})


def test_stream_timing_report(hip):
    """not a bar: microseconds per launch of both forms on the DiT's shape, back to back (printed)"""
    heads, S = 32, 4992
    qp, k, v = (t.cuda() for t in mk(1, S, S, heads, seed=1))
    res = {}
    for name, opt in (("stream", {"attn_q64_stream": "1"}), ("grid", {"attn_q64_stream": "0"}), ("stream2", {"attn_q64_stream": "1"}), ("grid2", {"attn_q64_stream": "0"})):
        with hip.options(**opt):
            for _ in range(20): hip.ops.attention_prescaled(qp, k, v, heads)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): hip.ops.attention_prescaled(qp, k, v, heads)
            e1.record(); torch.cuda.synchronize()
            res[name] = round(e0.elapsed_time(e1) / 200 * 1e3, 1)
    print({"us_per_launch": res, "TFLOPs": {n: round(4 * 32 * S * S * 64 / (us * 1e-6) / 1e12) for n, us in res.items()}})
