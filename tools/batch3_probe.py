#!/usr/bin/env python3
"""Experiment (round 5): the three guidance forwards of a C3 step (uncond / text / perturbed, t2v_pipeline.rs:860-940: three B = 1
calls in the reference) as ONE forward of batch 3 against three of batch 1.  Prints one JSON line (ms, GPU events, per step)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, ROOT)
import torch
import ltxhip
from ltxhip import schema
from bench import synth_on_device


def main():
    dev = "cuda:0"
    pre = ltxhip.get_config_by_version("0.9.5")
    F, H, W = 13, 16, 24
    S = F * H * W
    dit = ltxhip.LtxVideoTransformer3DModel(pre.transformer, synth_on_device(schema.dit_weight_shapes(pre.transformer), dev, 1), torch.bfloat16, 0)
    L = pre.transformer.num_layers
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(1, S, 128, generator=g).to(dev)
    pe = torch.randn(1, 128, 4096, generator=g).to(dev); ne = torch.randn(1, 128, 4096, generator=g).to(dev)
    pm = torch.zeros(1, 128); pm[:, :32] = 1; nm = torch.zeros(1, 128); nm[:, :12] = 1
    pm, nm = pm.to(dev), nm.to(dev)
    slm1 = torch.zeros(L, 1); slm1[19] = 1
    lat3 = lat.repeat(3, 1, 1).contiguous(); e3 = torch.cat([ne, pe, pe]).contiguous(); m3 = torch.cat([nm, pm, pm]).contiguous()
    slm3 = torch.zeros(L, 3); slm3[19, 2] = 1
    t1, t3 = [500.0], [500.0] * 3

    def three():
        a = dit.forward(lat, ne, t1, nm, F, H, W)
        b = dit.forward(lat, pe, t1, pm, F, H, W)
        c = dit.forward(lat, pe, t1, pm, F, H, W, skip_layer_mask=slm1)
        return a, b, c

    def one():
        return dit.forward(lat3, e3, t3, m3, F, H, W, skip_layer_mask=slm3)

    def timed(fn, n=6):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    a, b, c = three(); o = one()
    same = [bool(torch.equal(o[i:i + 1], x)) for i, x in enumerate((a, b, c))]
    err = [float((o[i:i + 1].float() - x.float()).norm() / x.float().norm()) for i, x in enumerate((a, b, c))]
    res = {"three_b1_ms": [], "one_b3_ms": []}
    for _ in range(2):
        res["three_b1_ms"].append(round(timed(three), 2)); res["one_b3_ms"].append(round(timed(one), 2))
    print(json.dumps({"what": "one C3 guidance step of the 2B DiT at 4992 tokens, bf16: three B=1 forwards vs one B=3 forward", **res,
                      "rows_bit_identical": same, "rel_l2": err}))


if __name__ == "__main__":
    main()
