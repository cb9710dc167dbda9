#!/usr/bin/env python3
"""head_dim 128 self-attention at the 13B model's size (BASELINE C5: S = 17556, 32 heads x 128): time and TF/s of the
prescaled bf16 path, plus a check against an f32 torch reference on a small slice of the queries."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
S, H, D = int(os.environ.get("S", 17556)), 32, 128
g = torch.Generator(device="cuda").manual_seed(1)
q = (torch.randn(1, S, H * D, device="cuda", generator=g) * (D ** -0.5) * 1.4426950408889634).bfloat16()
k = torch.randn(1, S, H * D, device="cuda", generator=g).bfloat16(); v = torch.randn(1, S, H * D, device="cuda", generator=g).bfloat16()
fn = lambda: ltxhip.ops.attention_prescaled(q, k, v, H)
ms = min(timeit(fn, iters=5, warm=2) for _ in range(3))
o = fn()
# reference on 64 queries of 2 heads
qs = q[0, 1000:1064].float().view(64, H, D)[:, :2]; ks = k[0].float().view(S, H, D)[:, :2]; vs = v[0].float().view(S, H, D)[:, :2]
p = torch.softmax(torch.einsum("qhd,khd->hqk", qs, ks) * math.log(2.0), dim=-1)
ref = torch.einsum("hqk,khd->qhd", p, vs)
got = o[0, 1000:1064].float().view(64, H, D)[:, :2]
err = float((got - ref).norm() / ref.norm())
print(json.dumps({"S": S, "ms": round(ms, 3), "TF": round(4 * S * S * H * D / ms / 1e9), "rel_l2_vs_f32": round(err, 5)}))
