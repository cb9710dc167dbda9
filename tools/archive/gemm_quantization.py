#!/usr/bin/env python3
"""Does a partly filled last round cost a whole tile-time?  Time vs tile count around the 256-CU boundaries, tile forced."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
tile = sys.argv[1] if len(sys.argv) > 1 else "256x256"
os.environ["LTX_GEMM_TILE"] = tile; os.environ["LTX_GEMM_SPLITK"] = "0"
bm, bn = [int(x) for x in tile.split("x")]
M, K = 20 * bm, 2048
x = torch.randn(M, K, device="cuda").bfloat16()
for ntn in (8, 12, 13, 24, 25, 26, 37, 38, 39, 51, 52):
    N = ntn * bn
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    ms = min(timeit(lambda: ltxhip.ops.linear(x, w, None), iters=10, warm=3) for _ in range(3))
    tiles = 20 * ntn
    print(json.dumps({"tile": tile, "tiles": tiles, "rounds_256": round(tiles / 256, 3), "us": round(ms * 1e3, 1), "us_per_round_equiv": round(ms * 1e3 / (tiles / 256), 1),
                      "TFLOPs": round(2 * M * N * K / ms / 1e9, 1)}))
