#!/usr/bin/env python3
"""Per-shape kernel micro-benchmarks through the C ABI (GPU only).  Prints one JSON line per case.
Used to A/B kernel variants in ONE process on the shapes of BASELINE C2 (rule: same process, random data)."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch  # noqa: E402
import ltxhip  # noqa: E402

dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm_cases():
    S = 4992
    for name, M, N, K, epi in [("qkv", S, 6144, 2048, 0), ("to_out", S, 2048, 2048, 2), ("q2", S, 2048, 2048, 0),
                               ("ff1", S, 8192, 2048, 1), ("ff2", S, 2048, 8192, 2), ("kv2", 128, 4096, 2048, 0),
                               ("proj_out", S, 128, 2048, 0), ("sq4096", 4096, 4096, 4096, 0)]:
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
        r = torch.randn(M, N, device=dev).bfloat16(); g = torch.randn(1, N, device=dev)
        fn = lambda: ltxhip.ops.linear(x, w, b, epi=epi, resid=r if epi >= 2 else None, gate=g if epi == 2 else None, rows_per_batch=M)
        ms = timeit(fn)
        print(json.dumps({"op": "gemm", "case": name, "M": M, "N": N, "K": K, "ms": round(ms, 4), "TFLOPs": round(2 * M * N * K / ms / 1e9, 1)}), flush=True)


def attn_cases():
    for name, Sq, Sk, H, hd, biased in [("self", 4992, 4992, 32, 64, False), ("cross", 4992, 128, 32, 64, True), ("self128", 4992, 4992, 16, 128, False)]:
        q = torch.randn(1, Sq, H * hd, device=dev).bfloat16(); k = torch.randn(1, Sk, H * hd, device=dev).bfloat16(); v = torch.randn(1, Sk, H * hd, device=dev).bfloat16()
        bias = torch.zeros(1, Sk, device=dev) if biased else None
        ms = timeit(lambda: ltxhip.ops.attention(q, k, v, H, 1 / math.sqrt(hd), bias))
        print(json.dumps({"op": "attn", "case": name, "ms": round(ms, 4), "TFLOPs": round(4 * H * Sq * Sk * hd / ms / 1e9, 1)}), flush=True)


def conv_cases():
    for name, C, T, H, W in [("mid1024", 1024, 13, 16, 24), ("up0_512", 512, 25, 32, 48), ("up1_256", 256, 49, 64, 96), ("up2_128", 128, 97, 128, 192)]:
        x = torch.randn(1, T, H, W, C, device=dev).bfloat16()
        w = (torch.randn(C, C, 3, 3, 3, device=dev) / math.sqrt(27 * C)).bfloat16(); b = torch.randn(C, device=dev).bfloat16()
        ms = timeit(lambda: ltxhip.ops.conv3d(x, w, b), iters=5, warm=1)       # includes the per-call weight repack (small)
        print(json.dumps({"op": "conv3d", "case": name, "ms": round(ms, 4), "TFLOPs": round(54 * C * C * T * H * W / ms / 1e9, 1)}), flush=True)


def norm_cases():
    for name, rows, D in [("dit_rms", 4992, 2048), ("vae_128", 97 * 128 * 192, 128), ("vae_256", 49 * 64 * 96, 256), ("vae_1024", 4992, 1024)]:
        x = torch.randn(rows, D, device=dev).bfloat16(); sc = torch.randn(1, D, device=dev); sh = torch.randn(1, D, device=dev)
        ms = timeit(lambda: ltxhip.ops.rownorm(x, 0, 1e-6, None, sc, sh, rows, 1))
        print(json.dumps({"op": "rownorm", "case": name, "ms": round(ms, 4), "GBs": round(2 * rows * D * 2 / ms / 1e6, 1)}), flush=True)
    rows, D = 4992, 2048
    x = torch.randn(rows, D, device=dev).bfloat16(); w = torch.ones(D, device=dev).bfloat16()
    c, s = ltxhip.ops.rope_table(1, 13, 16, 24, D, coords=ltxhip.build_video_coords(13, 16, 24).to(dev))
    ms = timeit(lambda: ltxhip.ops.qknorm_rope(x, w, 1e-5, c, s))
    print(json.dumps({"op": "qknorm_rope(+clone)", "ms": round(ms, 4)}), flush=True)


def tile_sweep():
    S = 4992
    for name, M, N, K in [("qkv", S, 6144, 2048), ("to_out", S, 2048, 2048), ("ff1", S, 8192, 2048), ("ff2", S, 2048, 8192), ("sq4096", 4096, 4096, 4096)]:
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
        res = {}
        for tile in ["256x256", "192x256", "128x256", "256x128", "192x128", "160x128", "128x128"]:
            with ltxhip.options(gemm_plan=tile):             # a gemm_big tile forced wherever the call is eligible for it
                ms = timeit(lambda: ltxhip.ops.linear(x, w, b))
            res[tile] = round(2 * M * N * K / ms / 1e9, 1)
        print(json.dumps({"op": "gemm_tile_sweep", "case": name, "TFLOPs": res}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "conv", "norm"]
    for w in which:
        {"gemm": gemm_cases, "attn": attn_cases, "conv": conv_cases, "norm": norm_cases, "tiles": tile_sweep}[w]()
