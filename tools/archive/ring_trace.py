#!/usr/bin/env python3
"""Phase stamps inside one gemm_ring launch (variant library built with -DRING_TRACE): per block, 100 MHz wall clock at
0 entry, 1 ring filled (NS stages issued), 2 first stage landed, 3 K loop done, 4 slab stored (split), 5 ticket drawn (split),
6 reduction done / epilogue starts, 7 outputs stored.  Prints mean segment lengths and the launch's span over all blocks."""
import ctypes, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LTXHIP_LIB"] = os.path.join(ROOT, "tools", "variants", "libltxhip_ring_trace.so")
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import numpy as np, torch, ltxhip
for name, M, N, K, epi, tile in [("qkv", 384, 6144, 2048, 0, "ring:96x96"), ("to_out", 384, 2048, 2048, 2, "ring:96x64"), ("ff1", 384, 8192, 2048, 1, "ring:96x128"), ("ff2", 384, 2048, 8192, 2, "ring:96x128"),
                                 ("t5_wi", 128, 10240, 4096, 1, "ring:128x96")]:
    os.environ["LTX_GEMM_RING_TILE"] = tile
    ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16() for _ in range(8)]
    x = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    resid = torch.randn(M, N, device="cuda").bfloat16(); gate = torch.randn(1, N, device="cuda")
    for i in range(12):
        ltxhip.ops.linear(x, ws[i % 8], b, epi=epi, resid=resid if epi == 2 else None, gate=gate if epi == 2 else None, rows_per_batch=M)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint32)
    assert ltxhip.lib.ltx_dbg_ring_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
    bm, bn = [int(v) for v in tile.split(":")[1].split("x")]
    nblk = -(-M // bm) * -(-N // bn) * int(os.environ.get("SF_" + name, "0") or 0) or None
    t = buf.reshape(1024, 8).astype(np.int64)
    live = t[:, 0] > 0
    t = t[live]
    split = (t[:, 4] >= t[:, 3]).all() and (t[:, 4] > 0).any()
    t0 = t[:, 0].min()
    seg = {"entry->ring_filled": (t[:, 1] - t[:, 0]).mean(), "ring_filled->first_stage": (t[:, 2] - t[:, 1]).mean(), "k_loop": (t[:, 3] - t[:, 2]).mean()}
    out = {"case": name, "tile": tile, "blocks": int(live.sum()), "launch_span_us": round(float(t[:, 1:].max() - t0) / 100, 2), "entry_skew_us": round(float(t[:, 0].max() - t0) / 100, 2)}
    if split:
        red = t[:, 6] >= t[:, 5]
        seg["slab_store"] = (t[:, 4] - t[:, 3]).mean(); seg["ticket"] = (t[:, 5] - t[:, 4]).mean()
        r = t[t[:, 7] >= t[:, 5]]
        seg["reduce(last arrivers)"] = (r[:, 6] - r[:, 5]).mean(); seg["epilogue(last arrivers)"] = (r[:, 7] - r[:, 6]).mean()
        out["last_arrivers"] = int(len(r))
    else:
        seg["drain+epilogue"] = (t[:, 7] - t[:, 3]).mean()
    out["mean_us"] = {k: round(float(v) / 100, 2) for k, v in seg.items()}
    print(json.dumps(out), flush=True)
