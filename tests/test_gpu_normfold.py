"""GPU suite: the DiT's RMS norm + AdaLN modulation folded into the GEMMs around it (round 6; GemmArgs::C2 / ::rs_sq, dit.hip).

LtxVideoTransformerBlock::forward (ltx_transformer.rs:847-851, 905-909) normalises h and modulates it before the q|k|v and ff1
projections: y = h * r * (1 + sc) + sh.  Times W^T that is r * ((h (.) (1 + sc)) W^T) + (sh W^T + b): the layer that writes h
also stores h (.) (1 + sc), the projection finishes with the row's 1 / rms (from the producer's row partials) and a per-timestep
vector.  bf16 production kernels only, where the partials come from gemm_asm16's epilogue (more than 512 rows, D <= 2048).
Bars: with the fold (default) and without (norm_fold=0) the forward stays within the bf16 bar against the oracle fed
bf16-rounded weights / inputs (rel-L2 <= 2e-2); the two arms differ from each other by rounding only (<= 1e-2), the fold's
distance to the oracle is not larger than the pass's by more than 10 %; the stand-alone norm launches are gone."""
import pytest
import torch

import ltx_oracle as O
from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFGD = dict(in_channels=128, out_channels=128, num_attention_heads=32, attention_head_dim=64, cross_attention_dim=2048, num_layers=3, caption_channels=4096)


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


def run(hip, model, args, **opts):
    with hip.options(**opts):
        hip.prof_enable(True)
        y = model.forward(*args)
        torch.cuda.synchronize()
        norms = hip.prof_report(4)[2]
        hip.prof_enable(False)
    return y.float().cpu(), norms


# (the row partials come from gemm_asm16's epilogue on the K = N = 2048 layers: more than 4096 rows in all - ltx_gemm_split_factor)
@pytest.mark.parametrize("B,F,H,W,skip", [(1, 13, 16, 24, None), (2, 6, 16, 26, None), (3, 4, 16, 26, "stg")])
def test_fold_matches_the_pass_and_the_oracle(hip, B, F, H, W, skip):
    cfg = O.DitConfig(**CFGD)
    w = O.synth_weights(O.dit_weight_shapes(cfg), seed=71)
    S, K = F * H * W, 128
    g = torch.Generator().manual_seed(72)
    hidden = torch.randn(B, S, 128, generator=g); enc = torch.randn(B, K, 4096, generator=g)
    mask = torch.zeros(B, K); mask[:, :40] = 1
    t = torch.tensor([896.0, 640.0, 100.0][:B])                 # exact in bf16; different rows see different modulation
    coords = O.build_video_coords(B, F, H, W)
    slm = None
    if skip == "stg":                                           # the guidance-batch form: row 2 skips block 1, rows 0 and 1 keep it
        slm = torch.zeros(3, B); slm[1, 2] = 1.0
    wr = {k: v.bfloat16().float() for k, v in w.items()}
    want = O.dit_forward(wr, cfg, hidden.bfloat16().float(), enc.bfloat16().float(), t, mask, F, H, W, None, coords, slm)
    model = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(**CFGD), {k: v.to(DEV) for k, v in w.items()}, torch.bfloat16)
    args = (hidden.to(DEV), enc.to(DEV), t, mask.to(DEV), F, H, W, None, coords.to(DEV), slm)
    y1, n1 = run(hip, model, args, norm_fold="1")
    y1b, _ = run(hip, model, args, norm_fold="1")
    y0, n0 = run(hip, model, args, norm_fold="0")
    assert torch.isfinite(y1).all() and torch.equal(y1, y1b)    # repeatable bit for bit
    e1, e0, d = rel_l2(y1, want), rel_l2(y0, want), rel_l2(y1, y0)
    print({"B": B, "S": S, "skip": skip, "fold_vs_oracle": round(e1, 5), "pass_vs_oracle": round(e0, 5), "fold_vs_pass": round(d, 5), "norm_launches": (n1, n0)})
    assert e1 <= 2e-2 and e0 <= 2e-2, (e1, e0)
    assert d <= 1e-2, d
    assert e1 <= 1.1 * e0 + 1e-4, (e1, e0)
    # per block two norms; with the fold only the first block's norm1 (nothing produced its rows) and the final LayerNorm remain
    # (+ none behind the blend: the restored rows' h (.) (1 + sc) is a map, not a norm)
    assert n0 == 2 * 3 + 1 and n1 == 2, (n1, n0)
    # norm_fold=2: the factor in per-timestep copies of the consumer's weights.  Rows at different timesteps (B = 2, 3 here) cannot
    # share a copy: the second-output form serves them - the bits of norm_fold=1
    y2, n2 = run(hip, model, args, norm_fold="2")
    if B == 1:
        e2 = rel_l2(y2, want)
        print({"weights_form_vs_oracle": round(e2, 5), "vs_second_output_form": round(rel_l2(y2, y1), 5)})
        assert n2 == 2 and e2 <= 1.1 * e0 + 1e-4 and rel_l2(y2, y1) <= 1e-2
        assert torch.equal(y2, run(hip, model, args, norm_fold="2")[0])
    else:
        assert torch.equal(y2, y1)


def test_weights_form_with_rows_at_one_timestep_and_a_cycling_schedule(hip):
    """norm_fold=2 on a guidance batch (three rows, ONE timestep, a row that skips a block) against the oracle; then more distinct
    timesteps than scaled-weight copies in a cycle: the handle gives the form up (it would re-scale 1.6 GB per step) and returns the
    second-output form's bits from then on."""
    cfg = O.DitConfig(**CFGD)
    w = O.synth_weights(O.dit_weight_shapes(cfg), seed=71)
    B, F, H, W, K = 3, 4, 16, 26, 128
    S = F * H * W
    g = torch.Generator().manual_seed(74)
    hidden = torch.randn(B, S, 128, generator=g); enc = torch.randn(B, K, 4096, generator=g)
    mask = torch.zeros(B, K); mask[:, :40] = 1
    coords = O.build_video_coords(B, F, H, W)
    slm = torch.zeros(3, B); slm[1, 2] = 1.0
    t = torch.tensor([896.0] * B)
    wr = {k: v.bfloat16().float() for k, v in w.items()}
    want = O.dit_forward(wr, cfg, hidden.bfloat16().float(), enc.bfloat16().float(), t, mask, F, H, W, None, coords, slm)
    model = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(**CFGD), {k: v.to(DEV) for k, v in w.items()}, torch.bfloat16)
    args = lambda tt: (hidden.to(DEV), enc.to(DEV), tt, mask.to(DEV), F, H, W, None, coords.to(DEV), slm)
    y2, n2 = run(hip, model, args(t), norm_fold="2")
    y0, _ = run(hip, model, args(t), norm_fold="0")
    e2, e0 = rel_l2(y2, want), rel_l2(y0, want)
    print({"weights_form_guidance_batch_vs_oracle": round(e2, 5), "pass": round(e0, 5)})
    assert n2 == 2 and e2 <= 1.1 * e0 + 1e-4
    # five timesteps cycling through two copies
    with hip.options(norm_fold="2", norm_fold_copies="2"):
        outs = {}
        for rnd in range(3):
            for tv in (896.0, 640.0, 384.0, 256.0, 128.0):
                outs[(rnd, tv)] = model.forward(*args(torch.tensor([tv] * B))).float().cpu()
    with hip.options(norm_fold="1"):
        ref = model.forward(*args(torch.tensor([128.0] * B))).float().cpu()
    assert torch.equal(outs[(2, 128.0)], ref)                  # given up by then: the second-output form
    assert torch.equal(outs[(1, 128.0)], outs[(2, 128.0)])
    # a long-unused copy is evicted and its buffer re-used (no thrash: each timestep runs nine forwards before the next one arrives)
    model2 = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(**CFGD), {k: v.to(DEV) for k, v in w.items()}, torch.bfloat16)
    with hip.options(norm_fold="2", norm_fold_copies="2"):
        seq = {}
        for tv in (896.0, 640.0, 384.0, 896.0):
            for _ in range(9):
                seq[tv] = model2.forward(*args(torch.tensor([tv] * B))).float().cpu()
    assert torch.equal(seq[896.0], y2)                          # still the scaled-weight form, rebuilt after its eviction: same bits


def test_guidance_rows_in_one_forward_keep_the_bits_of_separate_forwards(hip):
    """ltx_pipeline_call runs the guidance branches of a step as rows of one forward (pipeline.hip): a row that skips a layer must
    come out with the bits of a forward that never ran it - with the fold, the rows behind the blend take h (.) (1 + sc) from the
    same expression the epilogue uses."""
    cfg = O.DitConfig(**CFGD)
    w = O.synth_weights(O.dit_weight_shapes(cfg), seed=71)
    F, H, W, K = 13, 16, 24, 128                            # 4992 tokens: one-row and two-row forwards both take the fold (same kernels, same K partition)
    S = F * H * W
    g = torch.Generator().manual_seed(73)
    hidden = torch.randn(1, S, 128, generator=g); enc = torch.randn(1, K, 4096, generator=g)
    mask = torch.zeros(1, K); mask[:, :40] = 1
    coords = O.build_video_coords(1, F, H, W)
    model = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(**CFGD), {k: v.to(DEV) for k, v in w.items()}, torch.bfloat16)
    t1 = torch.tensor([896.0])
    one = lambda slm: model.forward(hidden.to(DEV), enc.to(DEV), t1, mask.to(DEV), F, H, W, None, coords.to(DEV), slm)
    plain = one(None)
    m1 = torch.zeros(3, 1); m1[1, 0] = 1.0
    pert = one(m1)
    slm = torch.zeros(3, 2); slm[1, 1] = 1.0
    both = model.forward(hidden.repeat(2, 1, 1).to(DEV), enc.repeat(2, 1, 1).to(DEV), torch.tensor([896.0, 896.0]), mask.repeat(2, 1).to(DEV), F, H, W, None,
                         coords.repeat(2, 1, 1).to(DEV), slm)
    torch.cuda.synchronize()
    assert torch.equal(both[0], plain[0]) and torch.equal(both[1], pert[0])
    assert not torch.equal(plain, pert)
