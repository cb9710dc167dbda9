#!/usr/bin/env python3
"""A/B a run-time OPTION of libltxhip (include/ltxhip.h, "run-time options"; value "-" = the option's default) on the C2 GEMM / conv /
attention shapes, interleaved rounds in ONE process.
usage: ab.py option=a,b [gemm|conv|attn|square ...]      e.g.  ab.py gemm_off=-,asm16 gemm"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
var, vals = sys.argv[1].split("="); vals = vals.split(",")
which = sys.argv[2:] or ["gemm", "conv"]
dev = "cuda"; S = 4992
cases = []
if "gemm" in which:
    for name, M, N, K, epi in [("qkv", S, 6144, 2048, 0), ("to_out", S, 2048, 2048, 2), ("ff1", S, 8192, 2048, 1), ("ff2", S, 2048, 8192, 2)]:
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
        r = torch.randn(M, N, device=dev).bfloat16(); g = torch.randn(1, N, device=dev)
        cases.append((name, 2 * M * N * K, (lambda x=x, w=w, b=b, r=r, g=g, epi=epi, M=M: ltxhip.ops.linear(x, w, b, epi=epi, resid=r if epi >= 2 else None, gate=g if epi == 2 else None, rows_per_batch=M))))
if "square" in which:      # the guide's reference shapes (one 256x256 tile per CU at 4096)
    for n in (4096, 8192):
        x = torch.randn(n, n, device=dev).bfloat16(); w = (torch.randn(n, n, device=dev) / math.sqrt(n)).bfloat16()
        cases.append((f"sq{n}", 2 * n ** 3, (lambda x=x, w=w: ltxhip.ops.linear(x, w, None))))
        if n == 4096:   # data dependence of the clock (DVFS): uniform [-1,1) and all-zero operands
            xu = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16(); wu = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16()
            cases.append((f"sq{n}_uniform", 2 * n ** 3, (lambda x=xu, w=wu: ltxhip.ops.linear(x, w, None))))
            xz = torch.zeros(n, n, device=dev).bfloat16(); wz = torch.zeros(n, n, device=dev).bfloat16()
            cases.append((f"sq{n}_zero", 2 * n ** 3, (lambda x=xz, w=wz: ltxhip.ops.linear(x, w, None))))
if "conv" in which:
    for name, C, T, H, W in [("mid1024", 1024, 13, 16, 24), ("up0_512", 512, 25, 32, 48), ("up1_256", 256, 49, 64, 96), ("up2_128", 128, 97, 128, 192)]:
        x = torch.randn(1, T, H, W, C, device=dev).bfloat16(); w = (torch.randn(C, C, 3, 3, 3, device=dev) / math.sqrt(27 * C)).bfloat16(); b = torch.randn(C, device=dev).bfloat16()
        cases.append((name, 54 * C * C * T * H * W, (lambda x=x, w=w, b=b: ltxhip.ops.conv3d(x, w, b))))
if "attn" in which:
    q, k, v = [torch.randn(1, S, 2048, device=dev).bfloat16() for _ in range(3)]
    cases.append(("attn_self", 4 * 32 * S * S * 64, (lambda q=q, k=k, v=v: ltxhip.ops.attention(q, k, v, 32, 0.125))))
    qp = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
    cases.append(("attn_self_prescaled", 4 * 32 * S * S * 64, (lambda qp=qp, k=k, v=v: ltxhip.ops.attention_prescaled(qp, k, v, 32))))
res = {c[0]: {v: [] for v in vals} for c in cases}
for rnd in range(3):
    for name, fl, fn in cases:
        for v in vals:
            ltxhip.set_option(var, None if v == "-" else v)
            ms = timeit(fn, iters=10, warm=2)
            res[name][v].append(fl / ms / 1e9)
ltxhip.set_option(var, None)
for name in res:
    print(json.dumps({"case": name, var: {v: round(sorted(res[name][v])[1], 1) for v in vals}}))
plans = {}
for name, M, N, K in [("qkv", S, 6144, 2048), ("to_out", S, 2048, 2048), ("ff1", S, 8192, 2048), ("ff2", S, 2048, 8192), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]:
    plans[name] = ltxhip.ops.gemm_plan(M, N, K)
for name, C, T, H, W in [("mid1024", 1024, 13, 16, 24), ("up0_512", 512, 25, 32, 48), ("up1_256", 256, 49, 64, 96), ("up2_128", 128, 97, 128, 192)]:
    plans[name] = ltxhip.ops.gemm_plan(T * H * W, C, C, 1, 27, T, H, W)
print(json.dumps({"tuned_plans": {k: v for k, v in plans.items() if v}}))
