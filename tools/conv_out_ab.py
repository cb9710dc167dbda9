#!/usr/bin/env python3
"""conv_out (128 -> 48 + unpatchify) at C2's size [97, 128, 192]: the 64-wide halo-staged tile against the per-tap 192 x 64 tile
(gemm_off=halo_out), interleaved rounds in one process; the op re-packs the weights per call (both arms pay it)."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
T, H, W, C = 97, 128, 192, 128
x = torch.randn(1, T, H, W, C, device="cuda").bfloat16(); w = (torch.randn(48, C, 3, 3, 3, device="cuda") / math.sqrt(27 * C)).bfloat16(); b = torch.randn(48, device="cuda").bfloat16()
t = {"per_tap": [], "halo64": []}
outs = {}
for rnd in range(4):
    for arm in (("per_tap", "halo64") if rnd % 2 == 0 else ("halo64", "per_tap")):
        ltxhip.set_option("gemm_off", "halo_out" if arm == "per_tap" else None)
        t[arm].append(timeit(lambda: ltxhip.ops.conv_out_unpatchify(x, w, b, postprocess=True), iters=5, warm=2))
        outs[arm] = ltxhip.ops.conv_out_unpatchify(x, w, b, postprocess=True)
fl = 54 * C * 48 * T * H * W
print(json.dumps({"per_tap_ms": round(min(t["per_tap"]), 3), "halo64_ms": round(min(t["halo64"]), 3), "per_tap_TF": round(fl / min(t["per_tap"]) / 1e9, 1),
                  "halo64_TF": round(fl / min(t["halo64"]) / 1e9, 1), "bit_identical": bool(torch.equal(outs["per_tap"], outs["halo64"]))}))
