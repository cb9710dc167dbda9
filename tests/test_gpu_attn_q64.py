"""GPU suite: the 64-queries-per-wave DiT self-attention kernel (csrc/attn_q64.hip; replaces candle-flash-attn at
ltx_transformer.rs:699-712) through the C ABI, against an f32 CPU softmax(q k^T) v on the same bf16 inputs.

Tolerance: rel-L2 <= 5e-3 (bf16 output rounding 2^-9 relative per element plus the bf16 rounding of P; measured
2.5e-3..3.5e-3), the bar VERDICT r1 asked for.  Covered: big (256-query) and small (128-query) blocks and their mix,
ragged query / key counts, every tile-count remainder of the unrolled-by-four main loop, batch > 1, 8 and 32 heads
(whole heads per XCD), scores growing past the first tile's maximum (P > 1 without a rescale), and the exact-max
fallback pass (a late key whose score exceeds the first tile's maximum by more than the f32 exponent range)."""
import math

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 5e-3
LOG2E = 1.4426950408889634


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


def ref_attn(qp, k, v, heads):
    """softmax over keys of ln2 * (q' k^T), f32, per head; q' is the prescaled bf16 query."""
    B, Sq, D = qp.shape
    hd = D // heads
    q4 = qp.float().view(B, Sq, heads, hd).transpose(1, 2).double()
    k4 = k.float().view(B, -1, heads, hd).transpose(1, 2).double()
    v4 = v.float().view(B, -1, heads, hd).transpose(1, 2).double()
    s = (q4 @ k4.transpose(-1, -2)) * math.log(2.0)
    p = torch.softmax(s, dim=-1)
    return (p @ v4).transpose(1, 2).reshape(B, Sq, D).float()


def mk(B, Sq, Sk, heads, seed, qscale=1.0):
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn(B, Sq, heads * 64, generator=g) * qscale).bfloat16()
    k = torch.randn(B, Sk, heads * 64, generator=g).bfloat16()
    v = torch.randn(B, Sk, heads * 64, generator=g).bfloat16()
    qp = (q.float() * (0.125 * LOG2E)).bfloat16()
    return qp, k, v


def run(hip, qp, k, v, heads):
    o = hip.ops.attention_prescaled(qp.cuda(), k.cuda(), v.cuda(), heads)
    torch.cuda.synchronize()
    o = o.float().cpu()
    assert torch.isfinite(o).all()
    return o


@pytest.mark.parametrize("B,Sq,Sk,heads", [
    (1, 256, 256, 8),        # one big block per head? (rounds rule -> small blocks), four full tiles
    (1, 4992, 4992, 8),      # C2 sequence length, 8 heads: big + small mix
    (1, 1000, 777, 8),       # ragged queries and keys
    (2, 300, 513, 8),        # batch 2, ragged, nine tiles
    (1, 130, 64, 3),         # heads not a multiple of 8 (plain block order), one tile
    (1, 64, 63, 2),          # a single ragged tile
    (1, 700, 128, 32),       # two tiles
    (1, 384, 384, 32),       # C1 geometry
])
def test_q64_vs_f32_reference(hip, B, Sq, Sk, heads):
    qp, k, v = mk(B, Sq, Sk, heads, seed=Sq + Sk)
    o = run(hip, qp, k, v, heads)
    ref = ref_attn(qp, k, v, heads)
    e = rel_l2(o, ref)
    assert e <= TOL, e


@pytest.mark.parametrize("Sk", [65, 128, 129, 192, 193, 256, 257, 320, 321, 384, 385, 448, 449, 512, 513, 576, 577, 640, 641, 704])
def test_q64_tile_counts(hip, Sk):
    """2..11 tiles, full and ragged last tile: every exit path of the unrolled-by-four loop, both block kinds."""
    heads = 8
    for Sq in (128, 2300):          # small blocks only / big blocks (8 per head = one round) + one small block
        qp, k, v = mk(1, Sq, Sk, heads, seed=Sk * 7 + Sq)
        o = run(hip, qp, k, v, heads)
        e = rel_l2(o, ref_attn(qp, k, v, heads))
        assert e <= TOL, (Sq, Sk, e)


def test_q64_forced_big_and_small_agree(hip, monkeypatch):
    """The big/small split is a speed choice: all-small, all-big and the default split give the same function."""
    heads, S = 8, 1536
    qp, k, v = mk(1, S, S, heads, seed=5)
    ref = ref_attn(qp, k, v, heads)
    outs = {}
    for nbig in ("0", "3", "6"):
        hip.set_option("attn_q64_big", nbig)
        outs[nbig] = run(hip, qp, k, v, heads)
        assert rel_l2(outs[nbig], ref) <= TOL, (nbig, rel_l2(outs[nbig], ref))
    hip.set_option("attn_q64_big", None)
    assert rel_l2(outs["0"], outs["6"]) <= 3e-3


def test_q64_scores_grow_past_first_tile(hip):
    """Fixed first-tile maximum: later keys score up to ~60 (log2 units) above it -> P up to 2^60, no rescale, no
    fallback; the result must still match (softmax is invariant to the subtracted constant)."""
    heads, Sq, Sk = 8, 520, 900
    qp, k, v = mk(1, Sq, Sk, heads, seed=11)
    kf = k.float()
    qf = qp.float()
    # make key 700 (tile 10) align with query 33 of every head strongly; key 333 moderately with query 400
    kf[0, 700] = qf[0, 33] * 4.0
    kf[0, 333] = qf[0, 400] * 2.0
    k = kf.bfloat16()
    o = run(hip, qp, k, v, heads)
    ref = ref_attn(qp, k, v, heads)
    assert rel_l2(o, ref) <= TOL, rel_l2(o, ref)
    # the rows that carry the spike, on their own
    for row in (33, 400):
        assert rel_l2(o[0, row], ref[0, row]) <= 2e-2, (row, rel_l2(o[0, row], ref[0, row]))


def test_q64_overflow_fallback(hip):
    """A late key scoring > 2^7 log2-units above the first tile's maximum overflows exp2 -> inf/NaN in l / O^T; the
    block must notice, compute exact row maxima and run again.  Checked against the f64 reference on every row, and
    against the generic kernel (running max)."""
    heads, Sq, Sk = 8, 300, 640
    qp, k, v = mk(1, Sq, Sk, heads, seed=21, qscale=2.0)
    qf, kf = qp.float(), k.float()
    kf[0, 600] = qf[0, 17] * 30.0          # score(q17, k600) = 30 |q'|^2 ~ 30 * 64 * (2*0.18)^2 ~ 250 per head
    kf[0, 5] = qf[0, 250] * 30.0           # first tile: dominates from the start (no overflow for this row)
    k = kf.bfloat16()
    s = (qp.float()[0, 17, :64] * k.float()[0, 600, :64]).sum()
    assert s > 160, float(s)                # the spike really is past the f32 exponent range from a tile-0 maximum of ~10
    o = run(hip, qp, k, v, heads)
    ref = ref_attn(qp, k, v, heads)
    assert rel_l2(o, ref) <= TOL, rel_l2(o, ref)
    assert rel_l2(o[0, 17], ref[0, 17]) <= 2e-2
    # twice in a row (the second pass must leave no state behind)
    o2 = run(hip, qp, k, v, heads)
    assert torch.equal(o, o2)


def test_q64_matches_previous_kernel(hip, monkeypatch):
    """Against the 32-query-wave pipelined kernel it replaces (running max, VALU row sums) on the C2 sequence length."""
    heads, S = 8, 4992
    qp, k, v = mk(1, S, S, heads, seed=3)
    o = run(hip, qp, k, v, heads)
    with hip.options(attn_off="q64"):
        o_old = run(hip, qp, k, v, heads)
    assert rel_l2(o, o_old) <= 4e-3, rel_l2(o, o_old)


def test_q64_deterministic_and_race_screen(hip):
    """LDS-DMA ring + counted vmcnt + one barrier per tile: repeated launches over odd sizes must be bit-identical."""
    g = torch.Generator().manual_seed(99)
    heads = 8
    for it in range(16):
        Sq = int(torch.randint(1, 1500, (1,), generator=g)); Sk = int(torch.randint(1, 1700, (1,), generator=g))
        qp, k, v = mk(1, Sq, Sk, heads, seed=1000 + it)
        outs = [run(hip, qp, k, v, heads) for _ in range(3)]
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), (Sq, Sk)
        e = rel_l2(outs[0], ref_attn(qp, k, v, heads))
        assert e <= TOL, (Sq, Sk, e)


# ---- persistent form (experiment builds, x_attn_q64_persist=1: static item lists, key-range parts merged in the launch): 32 heads, more
# blocks than CUs.  Not the default (measured 3 % slower than the block grid, attn_q64.hip), kept under test as an option.
def ref_attn_heads(qp, k, v, heads, pick):
    """f64 reference for the heads in `pick` only (the 32-head problems are too big to hold all score sets at once)."""
    B, Sq, D = qp.shape
    out = {}
    for hd_i in pick:
        sl = slice(hd_i * 64, hd_i * 64 + 64)
        s = (qp[..., sl].double() @ k[..., sl].double().transpose(-1, -2)) * math.log(2.0)
        out[hd_i] = (torch.softmax(s, dim=-1) @ v[..., sl].double()).float()
    return out


@pytest.mark.parametrize("B,Sq,Sk", [
    (1, 2304, 1024),         # 288 whole blocks on 256 CUs: every CU one part
    (1, 2500, 1536),         # ragged query count: 196-row last block (rows past Sq dropped in parts and in the merge)
    (1, 2400, 2048),         # 96-row last block -> small (128-query) blocks beside the parts
    (2, 1300, 1280),         # batch 2
    (1, 4992, 4992),         # the DiT launch itself
])
def test_q64_persistent_vs_f32_reference(hip, B, Sq, Sk, monkeypatch):
    if not hip.has_experiments(): pytest.skip("the persistent form is compiled into experiment builds only (make experiments)")
    hip.set_option("x_attn_q64_persist", "1")
    heads = 32
    qp, k, v = mk(B, Sq, Sk, heads, seed=Sq * 3 + Sk)
    o = run(hip, qp, k, v, heads)
    pick = (0, 7, 13, 24, 31)
    ref = ref_attn_heads(qp, k, v, heads, pick)
    for hd_i in pick:
        e = rel_l2(o[..., hd_i * 64:hd_i * 64 + 64], ref[hd_i])
        assert e <= TOL, (hd_i, e)


def test_q64_persistent_matches_block_grid_and_repeats(hip, monkeypatch):
    """Same function as the one-block-per-(head, 256 queries) grid (x_attn_q64_persist=0); which workgroup merges a split
    block depends on timing, the bits must not: ten back-to-back launches are identical."""
    heads, Sq, Sk = 32, 4992, 4992
    if not hip.has_experiments(): pytest.skip("the persistent form is compiled into experiment builds only (make experiments)")
    hip.set_option("x_attn_q64_persist", "1")
    qp, k, v = mk(1, Sq, Sk, heads, seed=77)
    qd, kd, vd = qp.cuda(), k.cuda(), v.cuda()
    outs = [hip.ops.attention_prescaled(qd, kd, vd, heads) for _ in range(10)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    hip.set_option("x_attn_q64_persist", "0")
    o_grid = hip.ops.attention_prescaled(qd, kd, vd, heads)
    torch.cuda.synchronize()
    e = rel_l2(outs[0].float().cpu(), o_grid.float().cpu())
    assert e <= 2e-3, e                     # split blocks sum their key ranges in a different order; whole blocks are bit-identical


def test_q64_persistent_parts_with_different_maxima(hip, monkeypatch):
    """A key in the LAST key range scores far above everything in the first: the parts carry different fixed maxima m_p and
    the merge rescales by 2^(m_p - M); a second spike overflows inside one part (exact-max pass of that part only)."""
    heads, Sq, Sk = 32, 2304, 1024
    if not hip.has_experiments(): pytest.skip("the persistent form is compiled into experiment builds only (make experiments)")
    hip.set_option("x_attn_q64_persist", "1")
    qp, k, v = mk(1, Sq, Sk, heads, seed=31, qscale=2.0)
    qf, kf = qp.float(), k.float()
    kf[0, 1000] = qf[0, 100] * 3.0          # late key (last part), moderate spike for query 100 of every head
    kf[0, 990] = qf[0, 1500] * 30.0         # late key, overflow-sized spike for query 1500 (tile 15 is never a part's first tile)
    kf[0, 3] = qf[0, 2000] * 30.0           # first tile of the first part: the dominant maximum from the start
    k = kf.bfloat16()
    o = run(hip, qp, k, v, heads)
    pick = (0, 5, 18, 31)
    ref = ref_attn_heads(qp, k, v, heads, pick)
    for hd_i in pick:
        sl = slice(hd_i * 64, hd_i * 64 + 64)
        assert rel_l2(o[..., sl], ref[hd_i]) <= TOL, (hd_i, rel_l2(o[..., sl], ref[hd_i]))
        for row in (100, 1500, 2000):
            assert rel_l2(o[0, row, sl], ref[hd_i][0, row]) <= 2e-2, (hd_i, row)
    o2 = run(hip, qp, k, v, heads)
    assert torch.equal(o, o2)
