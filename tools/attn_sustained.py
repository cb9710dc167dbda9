#!/usr/bin/env python3
"""Self-attention A/B under sustained load, on the DiT's own layout (q|k|v column slices of the fused [S, 6144] buffer).
usage: attn_sustained.py option=a,b      (a run-time option of include/ltxhip.h; "-" = its default), e.g. attn_q64_big=-,16"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
var, vals = sys.argv[1].split("="); vals = vals.split(",")
S = 4992
qkv = (torch.randn(1, S, 6144, device="cuda") * 0.5).bfloat16()
q, k, v = qkv[..., :2048], qkv[..., 2048:4096], qkv[..., 4096:]
sep = [t.contiguous() for t in (q, k, v)]
fn_f = lambda: ltxhip.ops.attention_prescaled(q, k, v, 32)
fn_s = lambda: ltxhip.ops.attention_prescaled(sep[0], sep[1], sep[2], 32)
timeit(fn_f, iters=300, warm=10)          # heat soak
res = {(lay, x): [] for lay in ("fused", "separate") for x in vals}
for rnd in range(5):
    for lay, fn in (("fused", fn_f), ("separate", fn_s)):
        for x in vals:
            ltxhip.set_option(var, None if x == "-" else x)
            res[(lay, x)].append(timeit(fn, iters=60, warm=3))
for (lay, x), ms in res.items():
    m = sorted(ms)[len(ms) // 2]
    print(json.dumps({"layout": lay, var: x, "us": round(m * 1e3, 1), "TFLOPs": round(4 * 32 * S * S * 64 / m / 1e9, 1)}))
