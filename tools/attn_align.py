#!/usr/bin/env python3
"""Is the d=64 self-attention time sensitive to where q, k, v, o sit relative to each other?  Dense [S, 2048] matrices carved
out of one arena at varying gaps."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
S, D = 4992, 2048
n = S * D
arena = torch.empty(4 * n + 64 * 1024 * 1024, dtype=torch.bfloat16, device="cuda")
src = (torch.randn(3, S, D, device="cuda") * 0.5).bfloat16()
warm = [torch.randn(1, S, D, device="cuda").bfloat16() for _ in range(3)]
timeit(lambda: ltxhip.ops.attention_prescaled(*warm, 32), iters=200, warm=5)
base = arena.data_ptr()
def carve(off_elems):
    return arena[off_elems:off_elems + n].view(1, S, D)
for gap_bytes in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, (2 << 20), (2 << 20) + 256):
    gap = gap_bytes // 2
    q = carve(0); k = carve(n + gap); v = carve(2 * (n + gap))
    q.copy_(src[0:1]); k.copy_(src[1:2]); v.copy_(src[2:3])
    ms = min(timeit(lambda: ltxhip.ops.attention_prescaled(q, k, v, 32), iters=30, warm=5) for _ in range(3))
    print(json.dumps({"gap_bytes": gap_bytes, "q_addr_mod_2M": base % (2 << 20), "us": round(ms * 1e3, 1), "TFLOPs": round(4 * 32 * S * S * 64 / ms / 1e9, 1)}))
