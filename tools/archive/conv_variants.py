#!/usr/bin/env python3
"""Times the VAE conv shapes with every tools/variants/libltxhip_halo*.so (HALO_ABL timing ablations of csrc/conv_halo.hip)."""
import json, math, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd"); VAR = os.path.join(ROOT, "tools", "variants")
if len(sys.argv) > 1 and sys.argv[1] == "measure":
    sys.path.insert(0, PKG); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch, ltxhip
    from microbench import timeit
    res = {}
    for name, C, T, H, W in [("up2_128", 128, 97, 128, 192), ("up1_256", 256, 49, 64, 96), ("up0_512", 512, 25, 32, 48)]:
        x = torch.randn(1, T, H, W, C, device="cuda").bfloat16(); w = (torch.randn(C, C, 3, 3, 3, device="cuda") / math.sqrt(27 * C)).bfloat16(); b = torch.randn(C, device="cuda").bfloat16()
        t = min(timeit(lambda: ltxhip.ops.conv3d(x, w, b), iters=5, warm=2) for _ in range(3))
        res[name] = {"ms": round(t, 3), "TF": round(54 * C * C * T * H * W / t / 1e9, 1), "plan": ltxhip.ops.gemm_plan(T * H * W, C, C, 1, 27, T, H, W)}
    print(json.dumps(res), flush=True)
else:
    for lib in sorted(f for f in os.listdir(VAR) if f.startswith("libltxhip_halo")):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "measure"], capture_output=True, text=True,
                           env=dict(os.environ, LTXHIP_LIB=os.path.join(VAR, lib), LTX_GEMM_TUNE="0", LTX_CONV_HALO="128"))
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(lib, line[-1] if line else ("FAILED " + p.stderr[-300:]), flush=True)
