#!/usr/bin/env python3
"""Self-attention fast path vs number of query blocks (grid rounds at 3 blocks per CU): Sk fixed at 4992."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
Sk = 4992
k, v = [torch.randn(1, Sk, 2048, device="cuda").bfloat16() for _ in range(2)]
for Sq in (1024, 2048, 3072, 3200, 4096, 4992, 6144, 9216):
    q = (torch.randn(1, Sq, 2048, device="cuda") * 0.18).bfloat16()
    ms = min(timeit(lambda: ltxhip.ops.attention_prescaled(q, k, v, 32), iters=10, warm=3) for _ in range(3))
    blocks = (Sq + 127) // 128 * 32
    print(json.dumps({"Sq": Sq, "blocks": blocks, "rounds_at_3_per_cu": round(blocks / 768, 3), "us": round(ms * 1e3, 1), "TFLOPs": round(4 * 32 * Sq * Sk * 64 / ms / 1e9, 1)}))
