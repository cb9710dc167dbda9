#!/usr/bin/env python3
"""Same-process A/B of the q/k RMS-norm + RoPE pass (qknorm_rope_fused_kernel) between the shipped library and a variant
(tools/variants/libltxhip_NAME.so): the DiT's launch, q and k of [4992, 2048] as two dense segments, through ltx_dit-free
plumbing: the op is reached with the C ABI's ltx_op_qknorm_rope per segment AND, for the two-segment launch the model makes,
by timing whole DiT forwards is not needed - the fused two-segment form is exposed here through a one-layer model's profile."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
libs = {"head": os.path.join(ROOT, "candle-video_amd", "libltxhip.so")}
for n in sys.argv[1:]: libs[n] = os.path.join(ROOT, "tools", "variants", f"libltxhip_{n}.so")
h = {k: C.CDLL(p, mode=os.RTLD_LOCAL | os.RTLD_NOW) for k, p in libs.items()}
S, D = 4992, 2048
x = torch.randn(S, D, device="cuda").bfloat16(); w = torch.randn(D, device="cuda").bfloat16()
cos = torch.rand(S, D // 2, device="cuda"); sin = torch.rand(S, D // 2, device="cuda")
vp = lambda t: C.c_void_p(t.data_ptr())
def launch(n):
    rc = h[n].ltx_op_qknorm_rope(vp(x), C.c_int64(S), D, D, vp(w), C.c_float(1e-5), vp(cos), vp(sin), 1, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
def timed(n, iters=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): launch(n)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for n in h: timed(n, 20)
res = {n: [] for n in h}
for r in range(6):
    for n in (list(h) if r % 2 == 0 else list(h)[::-1]): res[n].append(timed(n))
print(json.dumps({n: {"us_per_launch_one_segment": round(sorted(v)[len(v) // 2], 2), "all": [round(t, 2) for t in v]} for n, v in res.items()}))
