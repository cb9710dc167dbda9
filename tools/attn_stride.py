#!/usr/bin/env python3
"""Self-attention fast path vs the row stride of q/k/v (elements): separate buffers with stride LD, or column slices of one buffer."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
S = 4992
def run(tag, q, k, v):
    fn = lambda: ltxhip.ops.attention_prescaled(q, k, v, 32)
    ms = min(timeit(fn, iters=30, warm=5) for _ in range(3))
    print(json.dumps({"case": tag, "us": round(ms * 1e3, 1), "TFLOPs": round(4 * 32 * S * S * 64 / ms / 1e9, 1)}))
timeit(lambda: ltxhip.ops.attention_prescaled(*[torch.randn(1, S, 2048, device="cuda").bfloat16() for _ in range(3)], 32), iters=100, warm=5)
for LD in (2048, 2112, 2176, 4096, 6144, 6208, 6272, 8192):
    bufs = [(torch.randn(1, S, LD, device="cuda") * 0.5).bfloat16() for _ in range(3)]
    run(f"separate ld={LD}", *[b[..., :2048] for b in bufs])
for LD in (6144, 6208, 6272, 6400, 8192):
    b = (torch.randn(1, S, LD, device="cuda") * 0.5).bfloat16()
    run(f"fused ld={LD}", b[..., :2048], b[..., 2048:4096], b[..., 4096:6144])
