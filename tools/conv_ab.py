#!/usr/bin/env python3
"""Times the VAE's conv shapes at C2 (and one C5-sized stage) through ops.conv3d; prints TFLOP/s per shape."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
res = {}
for name, C, T, H, W in [("mid1024", 1024, 13, 16, 24), ("up0_512", 512, 25, 32, 48), ("up1_256", 256, 49, 64, 96), ("up2_128", 128, 97, 128, 192),
                         ("c5_up1_256", 256, 81, 352, 608)]:
    x = torch.randn(1, T, H, W, C, device="cuda").bfloat16()
    w = (torch.randn(C, C, 3, 3, 3, device="cuda") / math.sqrt(27 * C)).bfloat16(); b = torch.randn(C, device="cuda").bfloat16()
    ms = min(timeit(lambda: ltxhip.ops.conv3d(x, w, b), iters=4, warm=2) for _ in range(2))
    res[name] = {"ms": round(ms, 3), "TF": round(54 * C * C * T * H * W / ms / 1e9), "plan": ltxhip.ops.gemm_plan(T * H * W, C, C, 1, 27, T, H, W)}
    for force in os.environ.get("CONV_AB_FORCE", "").split(","):          # e.g. CONV_AB_FORCE=128,256: the halo kernel at that tile width
        if not force: continue
        with ltxhip.options(gemm_plan="halo:" + force):
            ms = min(timeit(lambda: ltxhip.ops.conv3d(x, w, b), iters=4, warm=2) for _ in range(2))
        res[name]["halo" + force] = round(54 * C * C * T * H * W / ms / 1e9)
    del x
print(json.dumps(res))
