#!/usr/bin/env python3
"""Per-workgroup timeline of the STREAM self-attention launch (csrc/attn_q64.hip attn_q64_stream_kernel) from a diagnostic build
(tools/attn_q64_tune.py build trace -DQ64_TRACE=1 -DQ64_TR_MASK=19: stamps at item start and statement end + what the item is):
per item kind the time from its start to the next item's start, the finish-time spread over the chip, and a least-squares fit
us = a * tiles + b per kind - the numbers the host scheduler's cost constants (Q64S_*) are set from.  usage: attn_stream_trace.py LIB"""
import ctypes, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd")
shutil.copyfile(sys.argv[1], os.path.join(PKG, "libltxhip.so"))
sys.path.insert(0, PKG)
import numpy as np, torch, ltxhip
S, heads = 4992, 32
q, k, v = [torch.randn(1, S, heads * 64, device="cuda").bfloat16() for _ in range(3)]
qp = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
for _ in range(400): ltxhip.ops.attention_prescaled(qp, k, v, heads)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): ltxhip.ops.attention_prescaled(qp, k, v, heads)
e1.record(); torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(PKG, "libltxhip.so"))
n = 256 * 8 * 4
buf = (ctypes.c_ulonglong * n)()
assert lib.ltx_dbg_q64_trace(buf, n) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(256, 8, 4)
t0 = int(a[:, 0, 0].min())
rows = []          # (kind, tiles, part, nparts, start, stmt_end, next_start or nan)
fin = []
for wg in range(256):
    items = []
    for i in range(8):
        if a[wg, i, 0] == 0: break
        info = int(a[wg, i, 3])
        items.append(dict(kind=info >> 48, tiles=(info >> 32) & 0xffff, part=(info >> 16) & 0xffff, nparts=info & 0xffff,
                          start=(int(a[wg, i, 0]) - t0) / 100.0, end=(int(a[wg, i, 1]) - t0) / 100.0))
    for i, it in enumerate(items):
        it["next"] = items[i + 1]["start"] if i + 1 < len(items) else float("nan")
        it["pos"] = i
        rows.append(it)
    fin.append(items[-1]["end"])
fin = np.array(fin)
out = {"us_per_launch_events": round(e0.elapsed_time(e1) * 10, 1), "finish_us[min,p25,med,p75,max]": [round(float(np.percentile(fin, p)), 1) for p in (0, 25, 50, 75, 100)]}
for kind, name in ((3, "part"), (2, "whole"), (1, "small")):
    r = [x for x in rows if x["kind"] == kind]
    if not r: continue
    t = np.array([x["tiles"] for x in r], dtype=float); d = np.array([x["end"] - x["start"] for x in r]); g = np.array([x["next"] - x["end"] for x in r])
    fit = np.polyfit(t, d, 1).tolist() if len(set(t.tolist())) > 1 else [float("nan"), float(d.mean())]
    first = np.array([x["pos"] == 0 for x in r])
    out[name] = {"n": len(r), "tiles[min,max]": [int(t.min()), int(t.max())], "statement_us[min,med,max]": [round(float(np.percentile(d, p)), 1) for p in (0, 50, 100)],
                 "us_per_tile_fit": round(fit[0], 3), "constant_us_fit": round(fit[1], 2), "gap_to_next_us[med,max]": [round(float(np.nanmedian(g)), 1), round(float(np.nanmax(g)), 1)] if np.isfinite(g).any() else None,
                 "first_in_list": int(first.sum())}
print(json.dumps(out))
order = np.argsort(-fin)
for wg in list(order[:4]) + list(order[-3:]):
    its = [x for x in rows if False]
print(json.dumps({"slowest": [[(x["kind"], x["tiles"], round(x["start"], 1), round(x["end"], 1)) for x in rows if x is not None and x.get("wg", None) is None][:0]]}))
