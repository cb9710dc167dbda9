#!/usr/bin/env python3
"""A/B of the DiT self-attention launch (S = 4992, 32 x 64) in one process: persistent item lists vs the block grid."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4992
heads = 32
q, k, v = [torch.randn(1, S, heads * 64, device="cuda").bfloat16() for _ in range(3)]
qp = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
fn = lambda: ltxhip.ops.attention_prescaled(qp, k, v, heads)
res = {}
for rnd in range(3):
    for name, env in (("persist", "1"), ("grid", "0")):
        os.environ["LTX_ATTN_Q64_PERSIST"] = env
        us = min(timeit(fn, iters=40, warm=3) for _ in range(3)) * 1e3
        res.setdefault(name, []).append(round(us, 1))
for name in list(res): res[name + "_TF"] = round(4 * heads * S * S * 64 / min(res[name]) / 1e6, 1)
print(json.dumps(res))
