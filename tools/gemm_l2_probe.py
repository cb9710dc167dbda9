#!/usr/bin/env python3
"""Runs 8192^3 and the qkv shape once through the vendor library and once through gemm_big (for rocprofv3 --pmc passes:
FETCH_SIZE / TCC_HIT_sum / TCC_MISS_sum per kernel tell how much of the operand stream each kernel's tile order serves from L2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch, ltxhip
for M, N, K in [(8192, 8192, 8192), (4992, 6144, 2048)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    for _ in range(3): torch.nn.functional.linear(x, w)
    for _ in range(3): ltxhip.ops.linear(x, w, None)
    torch.cuda.synchronize()
