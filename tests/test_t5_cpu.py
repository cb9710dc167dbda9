"""CPU suite, part 7: the T5 encoder restatement of the oracle, pinned against Hugging Face transformers' T5EncoderModel
(the model candle-transformers ported and the reference wraps, text_encoder.rs:315-345) on shared random weights."""
import pytest
import torch

import ltx_oracle as O

transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("S", [5, 40, 200])
def test_t5_oracle_matches_hf_transformers(S):
    from transformers import T5Config, T5EncoderModel
    cfg = O.T5Config(vocab_size=120, d_model=48, d_kv=8, d_ff=96, num_layers=3, num_heads=6)
    hf = T5EncoderModel(T5Config(vocab_size=120, d_model=48, d_kv=8, d_ff=96, num_layers=3, num_heads=6, relative_attention_num_buckets=32,
                                 relative_attention_max_distance=128, feed_forward_proj="gated-gelu", dropout_rate=0.0, layer_norm_epsilon=1e-6)).eval()
    g = torch.Generator().manual_seed(1)
    sd = hf.state_dict()
    for k, v in sd.items():                                   # HF's default init makes attention ~uniform: use O(1)-signal weights
        if v.dim() == 2 and "relative_attention_bias" not in k:
            v.copy_(torch.randn(v.shape, generator=g) / v.shape[1] ** 0.5 * (4.0 if "SelfAttention.q" in k else 1.0))
        elif "relative_attention_bias" in k:
            v.copy_(torch.randn(v.shape, generator=g))
        else:
            v.copy_(1.0 + 0.1 * torch.randn(v.shape, generator=g))
    sd["encoder.embed_tokens.weight"].copy_(sd["shared.weight"])
    p = {k: v.clone() for k, v in sd.items() if k in O.t5_weight_shapes(cfg)}
    assert set(p) == set(O.t5_weight_shapes(cfg)) and all(tuple(p[k].shape) == s for k, s in O.t5_weight_shapes(cfg).items())
    ids = torch.randint(0, 120, (2, S), generator=g)
    with torch.no_grad():
        want = hf(input_ids=ids).last_hidden_state
    got = O.t5_encoder_forward(p, cfg, ids)
    assert got.shape == want.shape == (2, S, 48)
    assert (got - want).abs().max() <= 2e-5 * max(1.0, float(want.abs().max())), float((got - want).abs().max())


def test_t5_relative_position_buckets_known_values():
    # hand-checkable points of the bidirectional bucketing: 32 buckets -> 16 per sign, exact up to 7, log beyond, clamp at 15
    rel = torch.tensor([0, 1, 7, 8, 11, 12, 127, 128, 1000, -1, -7, -8, -128])
    got = O.t5_relative_position_bucket(rel, 32, 128).tolist()
    assert got[:3] == [0, 17, 23] and got[3] == 24 and got[6] == 31 and got[7] == 31 and got[8] == 31
    assert got[9] == 1 and got[10] == 7 and got[11] == 8 and got[12] == 15
