#!/usr/bin/env python3
"""Does the operands' row stride (a power of two for every DiT linear) cost L2 channel conflicts?  Same M, N, near-equal K."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
for env in ({"LTX_GEMM_ASM": "0"}, {"LTX_GEMM_ASM": "1", "LTX_GEMM_ASM_TILE": "asm256x256"}):
    os.environ.pop("LTX_GEMM_ASM", None); os.environ.pop("LTX_GEMM_ASM_TILE", None); os.environ.update(env)
    for M, N, K in [(4096, 4096, 4096), (4096, 4096, 4160), (4096, 4096, 4032), (4992, 6144, 2048), (4992, 6144, 2112), (4992, 2048, 8192), (4992, 2048, 8256),
                    (4096, 4096, 16384), (4096, 4096, 16448), (8192, 8192, 8192), (8192, 8192, 8256)]:
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
        fn = lambda: ltxhip.ops.linear(x, w, None)
        t = min(timeit(fn, iters=10, warm=2) for _ in range(3))
        print(json.dumps({"env": env, "MNK": [M, N, K], "TF": round(2 * M * N * K / t / 1e9, 1), "plan": ltxhip.gemm_plan_name(M, N, K) if hasattr(ltxhip, "gemm_plan_name") else None}), flush=True)
