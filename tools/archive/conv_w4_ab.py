#!/usr/bin/env python3
"""A/B of the one-wave-per-SIMD conv_halo form (LTX_CONV_HALO_W4=1) against the shipped two-waves-per-SIMD pipelined form on the VAE's
128-wide conv launches (weights packed once, the conv launched through ltx_vae-free plumbing: ops.conv3d re-packs per call, so the
timing here uses many iterations and subtracts nothing - both arms pay the same packing); bit-identity of the outputs is checked."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
os.environ["LTX_GEMM_TUNE"] = "0"; os.environ["LTX_CONV_HALO"] = "128"
res = {}
for name, C, T, H, W in [("up2_128", 128, 97, 128, 192), ("up1_256", 256, 49, 64, 96), ("up0_512", 512, 25, 32, 48), ("c5ish_128_ragged", 128, 9, 176, 304)]:
    x = torch.randn(1, T, H, W, C, device="cuda").bfloat16(); w = (torch.randn(C, C, 3, 3, 3, device="cuda") / math.sqrt(27 * C)).bfloat16(); b = torch.randn(C, device="cuda").bfloat16()
    r = torch.randn(1, T, H, W, C, device="cuda").bfloat16()
    out = {}
    for arm in ("0", "1"):
        os.environ["LTX_CONV_HALO_W4"] = arm
        y = ltxhip.ops.conv3d(x, w, b); y2 = ltxhip.ops.conv3d(x, w, b, resid=r)
        out[arm] = (y.clone(), y2.clone())
    same = bool(torch.equal(out["0"][0], out["1"][0]) and torch.equal(out["0"][1], out["1"][1]))
    t = {"0": [], "1": []}
    for rnd in range(4):
        for arm in (("0", "1") if rnd % 2 == 0 else ("1", "0")):
            os.environ["LTX_CONV_HALO_W4"] = arm
            t[arm].append(timeit(lambda: ltxhip.ops.conv3d(x, w, b), iters=6, warm=2))
    fl = 54 * C * C * T * H * W
    res[name] = {"bit_identical": same, "two_waves_TF": round(fl / min(t["0"]) / 1e9, 1), "one_wave_TF": round(fl / min(t["1"]) / 1e9, 1)}
    print(json.dumps({name: res[name]}), flush=True)
    del x, r, out
