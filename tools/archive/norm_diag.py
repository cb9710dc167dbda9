#!/usr/bin/env python3
"""What a streaming pass over [4992, 2048] bf16 costs in three thread->chunk layouts (variant library -DLTX_NORM_DIAG), beside torch's
elementwise kernel and the shipped row norms; run under rocprofv3 --kernel-trace --stats, the kernel names tell the arms apart."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LTXHIP_LIB"] = os.path.join(ROOT, "tools", "variants", "libltxhip_normdiag.so")
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch, ltxhip
S, D = 4992, 2048
x = torch.randn(S, D, device="cuda").bfloat16(); y = torch.empty_like(x)
sc = torch.randn(1, D, device="cuda"); sh = torch.randn(1, D, device="cuda"); rs = ltxhip.ops.rowsq(x)
f = ltxhip.lib.ltx_dbg_norm_diag
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n = S * D // 8
for var in (0, 1, 2):
    for _ in range(50): assert f(x.data_ptr(), y.data_ptr(), n, var, D // 8, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(x, y); y.zero_()
    for _ in range(var + 1): torch.cuda.synchronize()
for _ in range(50): torch.mul(x, 1.5, out=y)
for _ in range(50): ltxhip.ops.rownorm_presum(x, rs, 1e-6, None, sc, sh, S, 0)
torch.cuda.synchronize()
