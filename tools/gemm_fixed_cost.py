#!/usr/bin/env python3
"""Fixed (K-independent) cost of one GEMM launch on the DiT's N = 2048 shape: time vs K, per epilogue."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
M, N = 4992, 2048
for epi in (0, 3):
    row = {}
    for K in (128, 512, 1024, 2048, 4096, 8192):
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
        r = torch.randn(M, N, device="cuda").bfloat16()
        f = (lambda: ltxhip.ops.linear(x, w, b, epi=3, resid=r)) if epi == 3 else (lambda: ltxhip.ops.linear(x, w, b))
        f(); row[K] = round(1000 * min(timeit(f, iters=20, warm=3) for _ in range(3)), 1)
    print(json.dumps({"epi": epi, "us_by_K": row}))
