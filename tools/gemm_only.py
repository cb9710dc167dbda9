#!/usr/bin/env python3
"""Runs one linear GEMM shape a few times (target of counter passes). usage: gemm_only.py M N K"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch, ltxhip
M, N, K = [int(x) for x in sys.argv[1:4]]
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
for _ in range(8):
    ltxhip.ops.linear(x, w, None)
torch.cuda.synchronize()
print(ltxhip.ops.gemm_plan(M, N, K))
