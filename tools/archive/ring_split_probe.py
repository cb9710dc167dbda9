#!/usr/bin/env python3
"""Small-M linear layers: which K partition (1 / 2 / 4 / 8 ranges, LTX_GEMM_SPLIT_FORCE) and which gemm_ring tile is fastest per
shape - the data behind the shape-only split rule of ltx_gemm_split_factor for M <= 512.  Kernel time from the library's
per-launch events, weights rotated through 8 copies."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch, ltxhip
TILES = ["ring:96x64", "ring:96x96", "ring:96x128", "ring:64x64", "ring:64x128", "ring:128x64", "ring:128x128", "ring:128x96", "ring:96x32", "ring:128x32", "ring:64x32"]
CASES = [(384, "qkv", 6144, 2048, 0), (384, "to_out", 2048, 2048, 2), (384, "ff1", 8192, 2048, 1), (384, "ff2", 2048, 8192, 2),
         (128, "ctx_kv", 4096, 2048, 0), (128, "t5_q", 4096, 4096, 0), (128, "t5_wi", 10240, 4096, 1), (128, "t5_wo", 4096, 10240, 3),
         (128, "cap1", 2048, 4096, 1), (256, "qkv", 6144, 2048, 0), (256, "to_out", 2048, 2048, 2), (256, "ff2", 2048, 8192, 2), (512, "to_out", 2048, 2048, 2), (512, "ff2", 2048, 8192, 2)]
def run(fn, iters=32):
    for _ in range(6): fn()
    torch.cuda.synchronize(); ltxhip.prof_enable(True)
    for _ in range(iters): fn()
    tot = cnt = 0
    for k in range(len(ltxhip.PROF_KERNELS)):
        ms, _, c = ltxhip.prof_report_kernel(0, k); tot += ms; cnt += c
    ltxhip.prof_enable(False)
    return tot / max(cnt, 1) * 1e3
for M, name, N, K, epi in CASES:
    ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16() for _ in range(8)]
    x = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    resid = torch.randn(M, N, device="cuda").bfloat16(); gate = torch.randn(1, N, device="cuda")
    i = [0]
    def fn():
        w = ws[i[0] % 8]; i[0] += 1
        return ltxhip.ops.linear(x, w, b, epi=epi, resid=resid if epi in (2, 3) else None, gate=gate if epi == 2 else None, rows_per_batch=M)
    row = {"M": M, "case": name, "N": N, "K": K}
    for sf in (1, 2, 4, 8):
        if (K // 64) // sf < 4: continue
        os.environ["LTX_GEMM_SPLIT_FORCE"] = str(sf)
        res = {}
        for t in TILES:
            bm, bn = [int(v) for v in t.split(":")[1].split("x")]
            if -(-M // bm) * -(-N // bn) * sf > 600: continue
            os.environ["LTX_GEMM_RING_TILE"] = t
            res[t[5:]] = round(run(fn), 1)
        os.environ.pop("LTX_GEMM_RING_TILE", None)
        best = min(res, key=res.get) if res else None
        row[f"sf{sf}"] = {"best": best, "us": res.get(best), "all": res}
    del os.environ["LTX_GEMM_SPLIT_FORCE"]
    print(json.dumps(row), flush=True)
