import sys, os, json, math
sys.path.insert(0, "/root/repo/candle-video_amd"); sys.path.insert(0, "/root/repo/tools")
import torch, ltxhip
from microbench import timeit
res = {}
for name, n, k in [("sq8192", 8192, 8192), ("k16k", 4096, 16384), ("sq4096", 4096, 4096)]:
    for kind in ("rand", "zero"):
        x = (torch.randn(n, k, device="cuda") if kind == "rand" else torch.zeros(n, k, device="cuda")).bfloat16()
        w = ((torch.randn(n, k, device="cuda") / math.sqrt(k)) if kind == "rand" else torch.zeros(n, k, device="cuda")).bfloat16()
        for lab, fn in (("vendor", lambda: torch.nn.functional.linear(x, w)), ("gemm_big", lambda: ltxhip.ops.linear(x, w, None))):
            t = min(timeit(fn, iters=10, warm=3) for _ in range(3))
            res[f"{name}_{kind}_{lab}"] = round(2 * n * n * k / t / 1e9)
print(json.dumps(res))
