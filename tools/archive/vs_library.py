#!/usr/bin/env python3
"""Reference point for the hand-written GEMM: the vendor library (torch.matmul -> hipBLASLt/rocBLAS) on the same C2 shapes,
same data, same timing loop.  Measurement only: the library is not on the product path."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
S = 4992
for name, M, N, K in [("qkv", S, 6144, 2048), ("to_out", S, 2048, 2048), ("ff1", S, 8192, 2048), ("ff2", S, 2048, 8192), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    res = {}
    for rnd in range(3):
        res.setdefault("ltxhip", []).append(2 * M * N * K / timeit(lambda: ltxhip.ops.linear(x, w, b), iters=10, warm=3) / 1e9)
        res.setdefault("library", []).append(2 * M * N * K / timeit(lambda: torch.nn.functional.linear(x, w, b), iters=10, warm=3) / 1e9)
        if os.environ.get("VS_ASM"):                       # the opt-in one-wave-per-SIMD asm loop beside them (LTX_GEMM_ASM is read per launch)
            os.environ["LTX_GEMM_ASM"] = os.environ["VS_ASM"]
            res.setdefault("asm", []).append(2 * M * N * K / timeit(lambda: ltxhip.ops.linear(x, w, b), iters=10, warm=3) / 1e9)
            del os.environ["LTX_GEMM_ASM"]
    print(json.dumps({"case": name, "TFLOPs": {k: round(sorted(v)[1], 1) for k, v in res.items()}, "plan": ltxhip.ops.gemm_plan(M, N, K)}))
