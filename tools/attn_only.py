#!/usr/bin/env python3
"""Runs the DiT self-attention (prescaled fast path, S=4992, 32x64) a few times; the target of counter passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch, ltxhip
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4992
q, k, v = [torch.randn(1, S, 2048, device="cuda").bfloat16() for _ in range(3)]
qp = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
for _ in range(6):
    ltxhip.ops.attention_prescaled(qp, k, v, 32)
torch.cuda.synchronize()
