#!/usr/bin/env python3
"""The four DiT linear layers at M = 384 on their gemm_ring tiles, 12 launches each with rotating weights - the workload of the
rocprofv3 --pmc passes behind DESIGN section 4's statement that these GEMMs are bound by L2 -> LDS traffic (TCC_REQ_sum x 128 B per
launch against (BM + BN) x K x 2 B x blocks), not by HBM (FETCH_SIZE x 2 against the weight bytes)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch, ltxhip
M = 384
for name, N, K, epi, tile in [("qkv", 6144, 2048, 0, "ring:96x96"), ("to_out", 2048, 2048, 2, "ring:96x32"), ("ff1", 8192, 2048, 1, "ring:96x128"), ("ff2", 2048, 8192, 2, "ring:96x32")]:
    os.environ["LTX_GEMM_RING_TILE"] = tile
    ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16() for _ in range(8)]
    x = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    resid = torch.randn(M, N, device="cuda").bfloat16(); gate = torch.randn(1, N, device="cuda")
    for i in range(12):
        ltxhip.ops.linear(x, ws[i % 8], b, epi=epi, resid=resid if epi == 2 else None, gate=gate if epi == 2 else None, rows_per_batch=M)
    torch.cuda.synchronize()
