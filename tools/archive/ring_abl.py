#!/usr/bin/env python3
"""Where a gemm_ring launch spends its time: the shipped kernel and its timing ablations (tools/ring_abl_build.sh), one child process
per library (LTXHIP_LIB), weights rotated through 8 copies (HBM / Infinity Cache) or held in one copy (L2-resident)."""
import json, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
    import torch, ltxhip
    out = {}
    for name, M, N, K, epi, tile in [("qkv", 384, 6144, 2048, 0, "ring:96x96"), ("to_out", 384, 2048, 2048, 0, "ring:96x64"), ("ff1", 384, 8192, 2048, 1, "ring:96x128"), ("ff2", 384, 2048, 8192, 0, "ring:96x128")]:
        os.environ["LTX_GEMM_RING_TILE"] = tile
        for copies in (8, 1):
            ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16() for _ in range(copies)]
            x = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
            i = [0]
            def fn():
                w = ws[i[0] % copies]; i[0] += 1
                return ltxhip.ops.linear(x, w, b, epi=epi)
            for _ in range(8): fn()
            torch.cuda.synchronize(); ltxhip.prof_enable(True)
            for _ in range(48): fn()
            ms, _, c = ltxhip.prof_report_kernel(0, ltxhip.PROF_KERNELS.index("gemm_ring_kernel")); ltxhip.prof_enable(False)
            out[f"{name}_{tile}_w{copies}"] = round(ms / max(c, 1) * 1e3, 2)
    print(json.dumps(out))
else:
    for abl in ["", "1", "2", "3", "4", "7"]:
        env = dict(os.environ)
        if abl: env["LTXHIP_LIB"] = os.path.join(ROOT, "tools", "variants", f"libltxhip_ring_a{abl}.so")
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
        print(json.dumps({"abl": abl or "none (shipped)", "what": {"": "", "1": "no MFMAs", "2": "no fragment reads", "3": "no MFMAs, no fragment reads", "4": "every piece re-reads one 1-KiB line set", "7": "all three"}[abl], "us": json.loads(line) if line.startswith("{") else line}), flush=True)
