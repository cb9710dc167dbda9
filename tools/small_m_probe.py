#!/usr/bin/env python3
"""Small-M linear layers (BASELINE config C1: 384 tokens; M = 128 text rows): time per launch against the time the weight
matrix alone needs at HBM rate - these GEMMs are weight-streaming problems, not MFMA problems."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
for M in [int(v) for v in os.environ.get("SMALLM_MS", "384,768,1152,128").split(",")]:
    for name, N, K, epi in [("qkv", 6144, 2048, 0), ("to_out", 2048, 2048, 2), ("ff1", 8192, 2048, 1), ("ff2", 2048, 8192, 2)]:
        # rotate through 8 weight copies so that the weights come from HBM as in the model (28 layers x 117 MB >> the 256 MB cache)
        ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16() for _ in range(8)]
        x = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
        resid = torch.randn(M, N, device="cuda").bfloat16(); gate = torch.randn(1, N, device="cuda")
        i = [0]
        def fn():
            w = ws[i[0] % 8]; i[0] += 1
            return ltxhip.ops.linear(x, w, b, epi=epi, resid=resid if epi == 2 else None, gate=gate if epi == 2 else None, rows_per_batch=M)
        t = min(timeit(fn, iters=16, warm=8) for _ in range(3))
        wb = N * K * 2
        print(json.dumps({"M": M, "case": name, "us": round(t * 1e3, 1), "TF": round(2 * M * N * K / t / 1e9), "weight_GBps": round(wb / t / 1e6), "hbm_floor_us": round(wb / 6.0e6, 1),
                          "plan": ltxhip.ops.gemm_plan(M, N, K)}), flush=True)
