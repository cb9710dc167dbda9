#!/usr/bin/env python3
"""DiT cross-attention (4992 queries x 128 text keys, 32 x 64, key bias) timing; env sweep of the block grouping."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
S, Sk = 4992, 128
q = torch.randn(1, S, 2048, device="cuda").bfloat16()
kv = torch.randn(1, Sk, 4096, device="cuda").bfloat16()
k, v = kv[..., :2048], kv[..., 2048:]
bias = torch.zeros(1, Sk, device="cuda"); bias[:, 100:] = -10000.0
fn = lambda: ltxhip.ops.attention(q, k, v, 32, 0.125, bias)
def t(tag):
    ms = min(timeit(fn, iters=50, warm=5) for _ in range(3))
    print(json.dumps({"case": tag, "us": round(ms * 1e3, 1)}))
os.environ["LTX_ATTN_CROSS"] = "0"; t("generic kernel")
os.environ["LTX_ATTN_CROSS"] = "1"
for g in (2, 4, 8, 12, 16, 20, 24, 39):
    os.environ["LTX_ATTN_CROSS_GROUPS"] = str(g); t(f"cross kernel groups={g}")
