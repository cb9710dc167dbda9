"""GPU suite: BASELINE config 5 as a MODEL (0.9.8-13B-distilled: D = 4096, 32 heads x 128, 48 layers, skip block 42;
704 x 1216 x 161 -> latent 21 x 22 x 38, S = 17556), at its full size.

The host oracle cannot run this size (one f32 score matrix set is 39 GB; a forward is 640 TFLOP), so the model is held to
size-independent properties instead, with the f32 parity mode of the same engine as the arithmetic reference - that mode's
kernels are the ones the oracle pins to <= 1e-3 at small sizes (test_gpu_models.py, test_gpu_c1.py):
  * one DiT forward, bf16 production kernels (gemm_big, attn_pipe128 / DMA attention, fused norms) vs f32 mode on the
    same bf16-representable weights and inputs: rel-L2 <= 2e-2 (C2 measures 0.5e-2 .. 0.7e-2 over 28 layers; 48 layers
    of the same rounding), no non-finite value;
  * skip_block_list = [42] (configs.rs:243-262) is honoured: the f32 mode runs with the same list, and the forward
    WITHOUT the list differs from the one with it by far more than the rounding distance;
  * two forwards of the same inputs are bit-identical (no run-to-run dependence at this size either).
"""
import math
import os
import sys

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c5_13b_full_forward_bf16_vs_f32_mode_and_skip_block():
    import ltxhip
    from ltxhip import schema
    sys.path.insert(0, ROOT)
    from bench import synth_on_device
    dev = "cuda:0"
    pre = ltxhip.get_config_by_version("0.9.8-13b-distilled")
    call = pre.pipeline_call(704, 1216, 161, postprocess=True)
    assert list(call.skip_block_list) == [42] and pre.transformer.num_layers == 48 and pre.transformer.attention_head_dim == 128
    F, H, W = 21, 22, 38
    S = F * H * W
    x = ltxhip.pack_latents(ltxhip.pcg32_randn(42, (1, 128, F, H, W))).to(dev)
    enc = torch.randn(1, 128, 4096, generator=torch.Generator().manual_seed(42)).bfloat16().float().to(dev)
    mask = torch.zeros(1, 128); mask[:, :32] = 1; mask = mask.to(dev)
    coords = ltxhip.build_video_coords(F, H, W)[None].to(dev)
    w = synth_on_device(schema.dit_weight_shapes(pre.transformer), dev, 1)          # bf16-representable matrices
    outs = {}
    for name, dt, skip in (("bf16", torch.bfloat16, [42]), ("bf16_noskip", torch.bfloat16, []), ("f32", torch.float32, [42])):
        m = ltxhip.LtxVideoTransformer3DModel(pre.transformer, w, dt, 0)
        m.set_skip_block_list(skip)
        y = m.forward(x, enc, [980.0], mask, F, H, W, video_coords=coords)
        if name == "bf16":
            y2 = m.forward(x, enc, [980.0], mask, F, H, W, video_coords=coords)
            assert torch.equal(y.view(torch.int16), y2.view(torch.int16))
        outs[name] = y.float().cpu()
        del m
        torch.cuda.empty_cache()
    assert outs["bf16"].shape == (1, S, 128)
    for v in outs.values():
        assert torch.isfinite(v).all()
    e = rel_l2(outs["bf16"], outs["f32"])
    assert e <= 2e-2, e
    d = rel_l2(outs["bf16_noskip"], outs["bf16"])
    assert d > 5 * e, (d, e)                        # block 42 is really dropped (the difference is far above rounding)
    assert 0.05 < float(outs["f32"].std()) < 50.0    # not degenerate
    print({"c5_forward_rel_l2_bf16_vs_f32": round(e, 5), "skip42_distance": round(d, 4), "out_std": round(float(outs["f32"].std()), 3)})
