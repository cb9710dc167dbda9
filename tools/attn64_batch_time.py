import sys, os, json, math
sys.path.insert(0, "/root/repo/candle-video_amd"); sys.path.insert(0, "/root/repo/tools")
import torch, ltxhip
from microbench import timeit
S, H, D = 4992, 32, 64
res = {}
for B in (1, 2, 3):
    g = torch.Generator(device="cuda").manual_seed(B)
    q = (torch.randn(B, S, H * D, device="cuda", generator=g) * 0.18).bfloat16(); k = torch.randn(B, S, H * D, device="cuda", generator=g).bfloat16(); v = torch.randn(B, S, H * D, device="cuda", generator=g).bfloat16()
    for tag, env in (("big16", "16"), ("new", None), ("big16b", "16"), ("newb", None)):
        ltxhip.set_option("attn_q64_big", env)                # None: the option's default (the greedy split)
        ms = min(timeit(lambda: ltxhip.ops.attention_prescaled(q, k, v, H), iters=10, warm=3) for _ in range(3))
        res[f"B{B}_{tag}"] = round(ms * 1000, 1)
    ltxhip.set_option("attn_q64_big", None)
print(json.dumps(res))
