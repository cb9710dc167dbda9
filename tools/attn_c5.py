#!/usr/bin/env python3
"""Self-attention at the 13B geometry (BASELINE config 5: S = 17556, 32 heads x 128): generic vs q-prescaled kernel."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
S, H, hd = 17556, 32, 128
q, k, v = [torch.randn(1, S, H * hd, device="cuda").bfloat16() for _ in range(3)]
qp = (q.float() * (hd ** -0.5 * 1.4426950408889634)).bfloat16()
fl = 4 * H * S * S * hd
for name, f in (("generic", lambda: ltxhip.ops.attention(q, k, v, H, hd ** -0.5)), ("prescaled", lambda: ltxhip.ops.attention_prescaled(qp, k, v, H))):
    ms = min(timeit(f, iters=5, warm=2) for _ in range(3))
    print(json.dumps({"case": "attn_c5_" + name, "ms": round(ms, 2), "TFLOPs": round(fl / ms / 1e9, 1)}))
