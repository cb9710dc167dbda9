"""GPU suite: the reference's DEFAULT text encoder, `QuantizedT5EncoderModel` over a GGUF file
(quantized_t5_encoder.rs:558-679; main.rs:441-444), through the C ABI.

  * ltx_gguf_dequantize (QTensor::dequantize on the device) against the oracle's dequantisers: the arithmetic is a handful
    of f32 products and one difference per element, each rounded separately on both sides, so the bar is BIT-EXACT f32
    (and bf16 = that f32 rounded once);
  * ltx_t5_create_from_gguf + ltx_t5_forward_masked on a small encoder whose tensors use the quantisations real T5 GGUF files
    mix (Q8_0, Q4_K, Q5_K, Q6_K, F16 and F32 norms) against the oracle forward on the oracle-dequantised weights, with the
    padding mask the reference passes: f32 mode <= 1e-3 rel-max (the path's f32 bar), bf16 within the bf16 model bar."""
import os
import sys

import numpy as np
import pytest
import torch

import ltx_oracle as O
from conftest import rel_l2, rel_max

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gguf_oracle as G


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


@pytest.mark.parametrize("name", ["F32", "F16", "BF16", "Q4_0", "Q5_0", "Q8_0", "Q4_K", "Q5_K", "Q6_K"])
def test_dequantize_kernel_bit_exact_vs_oracle(hip, name):
    ty = hip.GGML_TYPES[name]
    rng = np.random.default_rng(ty + 7)
    n = 256 * 37
    raw = G.random_blocks(ty, n, rng, scale=1.0)
    if name not in ("F32", "F16", "BF16"):                    # fully random payload bits too (every scale / bit-plane pattern)
        a = np.frombuffer(raw, np.uint8).reshape(-1, G.BLOCK[ty][1]).copy()
        keep = {"Q4_0": [(0, 2)], "Q5_0": [(0, 2)], "Q8_0": [(0, 2)], "Q4_K": [(0, 4)], "Q5_K": [(0, 4)], "Q6_K": [(208, 210)]}[name]
        b = rng.integers(0, 256, a.shape, dtype=np.uint8)
        for lo, hi in keep: b[:, lo:hi] = a[:, lo:hi]           # keep the half-precision scales finite
        raw = b.tobytes()
    want = torch.from_numpy(G.dequantize(ty, raw, n))
    got = hip.gguf_dequantize(ty, raw, n, torch.float32).cpu()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (got - want).abs().max()
    got_b = hip.gguf_dequantize(ty, raw, n, torch.bfloat16).cpu()
    assert torch.equal(got_b.view(torch.int16), want.bfloat16().view(torch.int16))


def test_dequantize_rejects_bad_arguments(hip):
    with pytest.raises(hip.LtxError):
        hip.gguf_dequantize(10, bytes(84), 256)               # Q2_K: not read
    with pytest.raises(hip.LtxError):
        hip.gguf_dequantize(hip.GGML_TYPES["Q8_0"], bytes(34), 31)


def small_t5_gguf(path, rng):
    cfg = O.T5Config(vocab_size=64, d_model=256, d_kv=64, d_ff=512, num_layers=2, num_heads=4)
    inner = cfg.num_heads * cfg.d_kv
    t = {}
    def add(name, ty, shape, scale):
        t[name] = (ty, shape, G.random_blocks(ty, int(np.prod(shape)), rng, scale))
    add("token_embd.weight", G.Q8_0, (cfg.vocab_size, cfg.d_model), 1.0)
    add("enc.blk.0.attn_rel_b.weight", G.F32, (cfg.relative_attention_num_buckets, cfg.num_heads), 0.5)
    kinds = [G.Q5_K, G.Q6_K, G.Q4_K, G.Q8_0, G.F16]
    k = 0
    for i in range(cfg.num_layers):
        p = f"enc.blk.{i}."
        for nm, shape, sc in [("attn_q", (inner, cfg.d_model), 0.06), ("attn_k", (inner, cfg.d_model), 0.06), ("attn_v", (inner, cfg.d_model), 0.06),
                              ("attn_o", (cfg.d_model, inner), 0.06), ("ffn_gate", (cfg.d_ff, cfg.d_model), 0.06), ("ffn_up", (cfg.d_ff, cfg.d_model), 0.06),
                              ("ffn_down", (cfg.d_model, cfg.d_ff), 0.04)]:
            add(p + nm + ".weight", kinds[k % len(kinds)], shape, sc); k += 1
        t[p + "attn_norm.weight"] = (G.F32, (cfg.d_model,), (1.0 + 0.1 * rng.standard_normal(cfg.d_model)).astype(np.float32).tobytes())
        t[p + "ffn_norm.weight"] = (G.F32, (cfg.d_model,), (1.0 + 0.1 * rng.standard_normal(cfg.d_model)).astype(np.float32).tobytes())
    t["enc.output_norm.weight"] = (G.F32, (cfg.d_model,), (1.0 + 0.1 * rng.standard_normal(cfg.d_model)).astype(np.float32).tobytes())
    G.write_gguf(path, t)
    names = G.t5_gguf_names(cfg.num_layers)
    w = {names[n]: torch.from_numpy(G.dequantize(ty, raw, int(np.prod(shape)))).reshape(shape) for n, (ty, shape, raw) in t.items()}
    return cfg, w


def test_quantized_t5_encoder_from_gguf_with_mask_vs_oracle(hip, tmp_path):
    path = str(tmp_path / "t5-small-mixed.gguf")
    cfg, w = small_t5_gguf(path, np.random.default_rng(5))
    B, S = 2, 40
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1, cfg.vocab_size, (B, S), generator=g)
    mask = torch.ones(B, S); mask[0, 23:] = 0; mask[1, 31:] = 0
    ids = ids * mask.long()                                      # pad id 0, as the tokenizer pads (text_encoder.rs:612-640)
    want = O.t5_encoder_forward(w, cfg, ids, torch.float32, attention_mask=mask)
    want_nomask = O.t5_encoder_forward(w, cfg, ids, torch.float32)
    hcfg = hip.T5EncoderConfig(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff, num_layers=cfg.num_layers, num_heads=cfg.num_heads)
    m = hip.T5TextEncoder.from_gguf(path, hcfg, torch.float32)
    got = m.forward(ids, mask).float().cpu()
    assert torch.isfinite(got).all()
    assert rel_max(got, want) <= 1e-3, rel_max(got, want)
    got_nm = m.forward(ids).float().cpu()
    assert rel_max(got_nm, want_nomask) <= 1e-3, rel_max(got_nm, want_nomask)
    assert rel_l2(want, want_nomask) > 1e-2                      # the mask matters on this input
    # the kept positions do not depend on what sits in the padded ones (the -1e9 bias removes them exactly)
    ids2 = ids.clone(); ids2[0, 23:] = 7; ids2[1, 31:] = 9
    got2 = m.forward(ids2, mask).float().cpu()
    assert torch.equal(got2[0, :23], got[0, :23]) and torch.equal(got2[1, :31], got[1, :31])
    del m
    mb = hip.T5TextEncoder.from_gguf(path, hcfg, torch.bfloat16)
    got_b = mb.forward(ids, mask).float().cpu()
    # the bf16 bar of the other models: no further from the f32 oracle than twice the oracle's own per-op-bf16 evaluation is
    d_ref = rel_l2(O.t5_encoder_forward(w, cfg, ids, torch.bfloat16, attention_mask=mask).float(), want)
    assert rel_l2(got_b, want) <= max(2.0 * d_ref, 2e-2), (rel_l2(got_b, want), d_ref)


def test_gguf_constructor_reports_missing_and_misshapen_tensors(hip, tmp_path):
    path = str(tmp_path / "t5.gguf")
    cfg, _ = small_t5_gguf(path, np.random.default_rng(6))
    t = G.read_gguf(path)
    hcfg = hip.T5EncoderConfig(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff, num_layers=cfg.num_layers, num_heads=cfg.num_heads)
    t2 = dict(t); del t2["enc.blk.1.ffn_up.weight"]
    G.write_gguf(path, t2)
    with pytest.raises(hip.LtxError, match="ffn_up"):
        hip.T5TextEncoder.from_gguf(path, hcfg, torch.float32)
    t3 = dict(t); ty, shape, raw = t3["enc.blk.0.attn_o.weight"]; t3["enc.blk.0.attn_o.weight"] = (ty, (shape[1], shape[0]) if shape[0] != shape[1] else (shape[0] // 2, shape[1] * 2), raw)
    G.write_gguf(path, t3)
    with pytest.raises(hip.LtxError, match="attn_o"):
        hip.T5TextEncoder.from_gguf(path, hcfg, torch.float32)
    with pytest.raises(hip.LtxError):
        hip.T5TextEncoder.from_gguf(str(tmp_path / "absent.gguf"), hcfg, torch.float32)
