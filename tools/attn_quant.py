#!/usr/bin/env python3
"""How much of the self-attention time is grid quantisation?  Same Sk, head count and kernel; Sq chosen so the block
count is / is not a whole number of rounds over the CU slots (3 blocks of 128 queries per CU -> 768 slots)."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
Sk, H = 4992, 32
k, v = [torch.randn(1, Sk, H * 64, device="cuda").bfloat16() for _ in range(2)]
for Sq in (1536, 3072, 4608, 4992, 6144):
    q = (torch.randn(1, Sq, H * 64, device="cuda") * 0.18).bfloat16()
    kk, vv = k, v
    # attention_prescaled needs Sq rows of q against Sk keys: q [1,Sq,D], k/v [1,Sk,D]
    ms = min(timeit(lambda: ltxhip.ops.attention_prescaled(q, kk, vv, H), iters=10, warm=2) for _ in range(3))
    blocks = math.ceil(Sq / 128) * H
    print(json.dumps({"Sq": Sq, "blocks": blocks, "rounds_at_768": round(blocks / 768, 3), "ms": round(ms, 4), "TFLOPs": round(4 * H * Sq * Sk * 64 / ms / 1e9, 1)}))
