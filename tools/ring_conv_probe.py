"""Small-plane 3x3x3 convs: every plan a shape may run, timed in one process (tools/archive/ab_round5/r5l_run.sh).
usage: python3 tools/ring_conv_probe.py  -> one JSON line per (shape, plan)"""
import json, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "candle-video_amd"))
import torch
import ltxhip as hip

SHAPES = [(1, 4, 8, 12, 1024, 1024, "C1 mid block"), (1, 4, 8, 12, 128, 1024, "C1 conv_in"), (1, 7, 16, 24, 512, 512, "C1 512-channel stage"),
          (16, 3, 4, 16, 1024, 1024, "C4 edge tiles, batch 16"), (1, 3, 16, 16, 1024, 1024, "one C4 leaf")]
PLANS = [None, "128x128", "192x256w16", "256x256w16", "ring:96x64", "ring:96x96", "ring:96x128", "ring:64x64", "ring:64x128", "ring:128x64", "ring:128x128", "ring:128x96"]


def timed(fn, n=30):
    """us per launch from the dispatch packets' own start / stop stamps (the call itself costs ~300 us of Python + ctypes)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    hip.prof_enable(True)
    for _ in range(n): fn()
    torch.cuda.synchronize()
    ms = cnt = 0
    for k in range(len(hip.PROF_KERNELS)):
        m, _, c = hip.prof_report_kernel(1, k); ms += m; cnt += c
    hip.prof_enable(False)
    return ms / max(cnt, 1) * 1e3


for B, T, H, W, Cin, Cout, what in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, H, W, Cin, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).bfloat16().cuda(); b = torch.randn(Cout, generator=g).bfloat16().cuda()
    r = torch.randn(B, T, H, W, Cout, generator=g).bfloat16().cuda()
    for plan in PLANS:
        try:
            if plan is None:
                us = timed(lambda: hip.ops.conv3d(x, w, b, False, resid=r)); name = "measured plan"
            else:
                with hip.options(gemm_plan=plan):
                    hip.prof_enable(True)
                    hip.ops.conv3d(x, w, b, False, resid=r)
                    cnt = hip.prof_report_kernel(1, hip.PROF_KERNELS.index("gemm_ring_kernel"))[2] if plan.startswith("ring") else 1
                    hip.prof_enable(False)
                    if not cnt: continue
                    us = timed(lambda: hip.ops.conv3d(x, w, b, False, resid=r)); name = plan
        except Exception as e:
            print(json.dumps({"shape": what, "plan": plan, "error": str(e)[:100]})); continue
        M = B * T * H * W
        print(json.dumps({"shape": what, "M": M, "Cin": Cin, "Cout": Cout, "plan": name, "us": round(us, 1), "TFLOPs": round(2 * 27 * M * Cin * Cout / us / 1e6, 1),
                          "weight_TBps": round(27 * Cin * Cout * 2 / us / 1e6, 2)}), flush=True)
