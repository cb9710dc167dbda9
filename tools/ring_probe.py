#!/usr/bin/env python3
"""Small-M linear layers: every gemm_ring tile against gemm_big's measured plan, kernel time from the library's per-launch events
(hipExtLaunchKernelGGL start/stop: no Python or launch overhead in the number), weights rotated through 8 copies so that they come
from HBM as in the model.  One JSON line per (shape, arm); `same_bits` compares with the gemm_big arm."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch, ltxhip
TILES = ["ring:96x64", "ring:96x96", "ring:96x128", "ring:64x64", "ring:64x128", "ring:128x64", "ring:128x128", "ring:128x96"]
SHAPES = [("qkv", 6144, 2048, 0), ("to_out", 2048, 2048, 2), ("ff1", 8192, 2048, 1), ("ff2", 2048, 8192, 2)]
T5 = [("t5_qkv", 4096, 4096, 0), ("t5_wi", 10240, 4096, 1), ("t5_wo", 4096, 10240, 3), ("ctx_kv", 4096, 2048, 0)]
KID = {n: i for i, n in enumerate(ltxhip.PROF_KERNELS)}
def run(fn, iters=48):
    for _ in range(8): fn()
    torch.cuda.synchronize(); ltxhip.prof_enable(True)
    for _ in range(iters): fn()
    tot = cnt = 0
    for k in range(len(ltxhip.PROF_KERNELS)):
        ms, _, c = ltxhip.prof_report_kernel(0, k); tot += ms; cnt += c
    ltxhip.prof_enable(False)
    return tot / max(cnt, 1) * 1e3, cnt
for M in [int(v) for v in os.environ.get("RING_MS", "384,128").split(",")]:
    for name, N, K, epi in (SHAPES if M != 128 else SHAPES + T5):
        ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16() for _ in range(8)]
        x = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
        resid = torch.randn(M, N, device="cuda").bfloat16(); gate = torch.randn(1, N, device="cuda")
        i = [0]
        def fn():
            w = ws[i[0] % 8]; i[0] += 1
            return ltxhip.ops.linear(x, w, b, epi=epi, resid=resid if epi in (2, 3) else None, gate=gate if epi == 2 else None, rows_per_batch=M)
        ltxhip.set_option("gemm_off", "ring")
        i[0] = 0; ref = fn().clone(); us, cnt = run(fn)
        print(json.dumps({"M": M, "case": name, "N": N, "K": K, "arm": "gemm_big (measured plan, ring family off)", "plan": ltxhip.ops.gemm_plan(M, N, K), "us": round(us, 2),
                          "TF": round(2 * M * N * K / us / 1e6), "weight_GBps": round(N * K * 2 / us / 1e3)}), flush=True)
        ltxhip.set_option("gemm_off", None)
        for t in TILES:
            ltxhip.set_option("gemm_plan", "ring:" + t)
            i[0] = 0; got = fn().clone(); us, cnt = run(fn)
            ltxhip.set_option("gemm_plan", None)
            print(json.dumps({"M": M, "case": name, "arm": t, "us": round(us, 2), "TF": round(2 * M * N * K / us / 1e6), "weight_GBps": round(N * K * 2 / us / 1e3),
                              "same_bits": bool(torch.equal(got.view(torch.int16), ref.view(torch.int16)))}), flush=True)
