#!/usr/bin/env python3
"""Timeline of the persistent self-attention launch from a -DQ64_TRACE=1 build (tools/attn_q64_tune.py build trace -DQ64_TRACE=1):
per workgroup and item the 10-ns stamps at item start / block done / published / merged.  usage: attn_q64_trace.py LIB [S]"""
import ctypes, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd")
lib_path = sys.argv[1]
shutil.copyfile(lib_path, os.path.join(PKG, "libltxhip.so"))
sys.path.insert(0, PKG)
import numpy as np, torch, ltxhip
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4992
heads = 32
q, k, v = [torch.randn(1, S, heads * 64, device="cuda").bfloat16() for _ in range(3)]
qp = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
for _ in range(400): ltxhip.ops.attention_prescaled(qp, k, v, heads)       # (clocks settled: the stamps of the LAST launch are read)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): ltxhip.ops.attention_prescaled(qp, k, v, heads)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"us_per_launch_events": round(e0.elapsed_time(e1) * 10, 1)}))
lib = ctypes.CDLL(os.path.join(PKG, "libltxhip.so"))
n = 256 * 8 * 4
buf = (ctypes.c_ulonglong * n)()
assert lib.ltx_dbg_q64_trace(buf, n) == 0
a = np.array(buf[:], dtype=np.float64).reshape(256, 8, 4)
t0 = a[:, 0, 0].min()
a = np.where(a > 0, (a - t0) / 100.0, np.nan)          # microseconds since the first workgroup started
def st(x): x = x[~np.isnan(x)]; return [round(float(np.percentile(x, p)), 2) for p in (0, 50, 100)] if len(x) else None
res = {"start_us[min,med,max]": st(a[:, 0, 0])}
for i in range(4):
    if np.isnan(a[:, i, 0]).all(): break
    res[f"item{i}"] = {"block": st(a[:, i, 1] - a[:, i, 0]), "publish": st(a[:, i, 2] - a[:, i, 1]), "merge": st(a[:, i, 3] - a[:, i, 2]),
                       "end_at": st(np.nanmax(a[:, i, :], axis=1))}
res["finish_us[min,med,max]"] = st(np.nanmax(a.reshape(256, -1), axis=1))
print(json.dumps(res))
# the ten slowest workgroups' timelines
fin = np.nanmax(a.reshape(256, -1), axis=1)
for wg in np.argsort(-fin)[:6]:
    print(int(wg), [[None if np.isnan(x) else round(float(x), 1) for x in a[wg, i]] for i in range(4)])
