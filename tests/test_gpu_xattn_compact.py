"""GPU suite: cross attention over the keys the text mask leaves alive (round 5).

The reference adds (1 - mask) * -10000 to the scores of every text key (ltx_transformer.rs:1059-1070) and runs a plain softmax
over all of them (:719-740).  A masked key's weight is exp(s - 10000 - max) = +0.0f exactly in f32 (scores would have to differ
by thousands for it not to be), so the engine moves the keys whose bias is above -5000 to the front of their batch row once per
context (key_compact_kernel + gather_rows_kernel, order kept) and attn_cross64_kernel multiplies ceil(count / 32) key blocks
instead of ceil(K / 32): BASELINE's prompts keep 32 of 128 tokens.  Checked here through the C ABI: the kept-key counts; the
compacted path against ltx_op_attention on the same inputs (prefix masks: the same bits; any mask: within a bf16 rounding of
the output) and against an f32 reference; every-key-masked rows (a constant shift: all keys kept); fractional mask values; the
folded q-norm on top; a whole DiT forward with a non-prefix mask against the oracle with compaction on and off."""
import pytest
import torch

import ltx_oracle as O
from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


def ref_attn(q, k, v, heads, scale, bias):
    B, Sq, D = q.shape
    hd = D // heads
    qh = q.float().view(B, Sq, heads, hd).transpose(1, 2); kh = k.float().view(B, -1, heads, hd).transpose(1, 2); vh = v.float().view(B, -1, heads, hd).transpose(1, 2)
    att = qh @ kh.transpose(-1, -2) * scale + bias[:, None, None, :]
    return (torch.softmax(att, -1) @ vh).transpose(1, 2).reshape(B, Sq, D)


def make(B, Sq, Sk, heads, seed):
    g = torch.Generator().manual_seed(seed)
    D = heads * 64
    q = torch.randn(B, Sq, D, generator=g).bfloat16().to(DEV)
    k = torch.randn(B, Sk, D, generator=g).bfloat16().to(DEV); v = torch.randn(B, Sk, D, generator=g).bfloat16().to(DEV)
    return q, k, v


@pytest.mark.parametrize("nvalid", [1, 5, 31, 32, 33, 64, 65, 100, 128])
def test_prefix_masks_same_bits_as_the_full_key_set(hip, nvalid):
    """BASELINE's mask shape (the first n tokens valid): compaction moves nothing, the kernel only stops after ceil(n / 32) key
    blocks; the dropped blocks contributed exact zeros, so the output bits are those of the full-key-set launch."""
    B, Sq, Sk, heads = 1, 4992 if nvalid == 32 else 777, 128, 32 if nvalid == 32 else 4
    q, k, v = make(B, Sq, Sk, heads, 100 + nvalid)
    bias = torch.zeros(B, Sk); bias[:, nvalid:] = -10000.0; bias = bias.to(DEV)
    full = hip.ops.attention(q, k, v, heads, 0.125, bias)
    comp, cnt = hip.ops.attention_compact(q, k, v, heads, 0.125, bias)
    assert cnt.tolist() == [nvalid]
    assert torch.equal(comp, full)
    assert rel_l2(comp.float().cpu(), ref_attn(q, k, v, heads, 0.125, bias).cpu()) <= 5e-3


@pytest.mark.parametrize("B,Sq,Sk,heads", [(2, 333, 128, 8), (3, 65, 77, 4), (1, 1000, 96, 2)])
def test_scattered_masks_per_batch_counts(hip, B, Sq, Sk, heads):
    """Non-prefix masks, a different number of valid keys per batch row (incl. one row with none valid and one with all): counts,
    the f32 reference at the bf16 bar, and the full-key-set launch to within one bf16 rounding of the largest output (a key's
    position inside the matrix instruction's k group changes, zeros between the products do not)."""
    g = torch.Generator().manual_seed(Sq)
    q, k, v = make(B, Sq, Sk, heads, Sq + Sk)
    keep = torch.rand(B, Sk, generator=g) < 0.3
    keep[0] = False                                             # every key masked: a constant shift, all keys kept
    if B > 2: keep[2] = True
    bias = torch.where(keep, torch.zeros(()), torch.full((), -10000.0)).to(DEV)
    full = hip.ops.attention(q, k, v, heads, 0.125, bias)
    comp, cnt = hip.ops.attention_compact(q, k, v, heads, 0.125, bias)
    want_cnt = [int(r.sum()) if r.any() else Sk for r in keep]
    assert cnt.tolist() == want_cnt
    ref = ref_attn(q, k, v, heads, 0.125, bias).cpu()
    e_c, e_f = rel_l2(comp.float().cpu(), ref), rel_l2(full.float().cpu(), ref)
    print(f"scattered mask B={B} Sk={Sk}: counts {want_cnt}; rel-L2 vs f32: compact {e_c:.5f}, full {e_f:.5f}")
    assert e_c <= 5e-3 and e_c <= 1.1 * e_f + 1e-5
    assert (comp.float() - full.float()).abs().max() <= 2 ** -7 * full.float().abs().max()
    assert torch.equal(comp[0], full[0])                        # the all-masked row keeps every key where it was


def test_fractional_mask_values_keep_their_bias(hip):
    """A mask value strictly between 0 and 1 gives a finite bias (0.9999 -> -1.0, 0.7 -> -3000): such keys stay, with their bias;
    only biases at or below -5000 are dropped."""
    B, Sq, Sk, heads = 1, 200, 128, 4
    q, k, v = make(B, Sq, Sk, heads, 7)
    bias = torch.full((B, Sk), -10000.0)
    bias[0, 3] = 0.0; bias[0, 17] = -1.0; bias[0, 64] = -2.5; bias[0, 90] = -3000.0; bias[0, 127] = -4999.0; bias[0, 100] = -5000.0
    bias = bias.to(DEV)
    comp, cnt = hip.ops.attention_compact(q, k, v, heads, 0.125, bias)
    assert cnt.tolist() == [5]
    ref = ref_attn(q, k, v, heads, 0.125, bias).cpu()
    assert rel_l2(comp.float().cpu(), ref) <= 5e-3
    full = hip.ops.attention(q, k, v, heads, 0.125, bias)
    assert (comp.float() - full.float()).abs().max() <= 2 ** -7 * full.float().abs().max()


def test_compaction_under_the_folded_q_norm(hip):
    """AttnArgs::q_rowsq and k_count together (the DiT's launch): un-normalised queries, 40 of 128 keys, vs the same launch on all keys."""
    B, Sq, Sk, heads = 2, 555, 128, 8
    q, k, v = make(B, Sq, Sk, heads, 9)
    D = heads * 64
    bias = torch.zeros(B, Sk); bias[0, 40:] = -10000.0; bias[1, ::2] = -10000.0; bias = bias.to(DEV)
    rs = hip.ops.rowsq(q.view(B * Sq, D))
    full = hip.ops.attention_rowsq(q, k, v, heads, 0.125, bias, rs, 1e-5)
    comp, cnt = hip.ops.attention_compact(q, k, v, heads, 0.125, bias, rs, 1e-5)
    assert cnt.tolist() == [40, 64]
    assert torch.equal(comp[0], full[0])
    assert (comp.float() - full.float()).abs().max() <= 2 ** -7 * full.float().abs().max()


def test_dit_forward_with_a_scattered_mask_vs_oracle(hip):
    """ltx_dit_forward (bf16, D = 512, 128 text tokens, batch of two with different scattered masks) with the compaction on
    (default) and off (option xattn_compact=0): both within the bf16 bar of the f32 oracle, and within rounding of each other."""
    cfgd = dict(in_channels=32, out_channels=32, num_attention_heads=8, attention_head_dim=64, cross_attention_dim=512, num_layers=2, caption_channels=64)
    cfg = O.DitConfig(**cfgd)
    w = O.synth_weights(O.dit_weight_shapes(cfg), seed=79)
    Fr, H, W, K = 2, 7, 9, 128
    S = Fr * H * W
    g = torch.Generator().manual_seed(80)
    hidden = torch.randn(2, S, 32, generator=g); enc = torch.randn(2, K, 64, generator=g)
    mask = (torch.rand(2, K, generator=g) < 0.25).float(); mask[1, 100:] = 1
    coords = O.build_video_coords(2, Fr, H, W)
    t = torch.tensor([896.0, 640.0])
    wr = {k: v.bfloat16().float() for k, v in w.items()}
    ref = O.dit_forward(wr, cfg, hidden.bfloat16().float(), enc.bfloat16().float(), t, mask, Fr, H, W, None, coords)
    outs = {}
    for tag, val in (("on", None), ("off", "0")):
        with hip.options(xattn_compact=val):
            model = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(**cfgd), {k: v.to(DEV) for k, v in w.items()}, torch.bfloat16)
            outs[tag] = model.forward(hidden.to(DEV), enc.to(DEV), t, mask.to(DEV), Fr, H, W, None, coords.to(DEV)).float().cpu()
    e_on, e_off = rel_l2(outs["on"], ref), rel_l2(outs["off"], ref)
    print(f"dit scattered mask bf16 vs f32 oracle: compact {e_on:.5f}, full {e_off:.5f}; between them {rel_l2(outs['on'], outs['off']):.6f}")
    assert e_on <= 2e-2 and e_on <= 1.25 * e_off + 1e-3
    assert rel_l2(outs["on"], outs["off"]) <= 5e-3
