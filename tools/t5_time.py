#!/usr/bin/env python3
"""T5-XXL encoder (24 layers, d_model 4096, 64 heads x 64, d_ff 10240; 4.7 B parameters) at the pipeline's 128 tokens:
time per prompt with random weights (the step before the denoise path; not part of the headline metric)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, ltxhip
import ltx_oracle as O
cfg = O.T5Config()
w = {}
for k, shp in O.t5_weight_shapes(cfg).items():
    if len(shp) == 2 and "relative_attention_bias" not in k:
        w[k] = (torch.randn(shp, device="cuda", dtype=torch.bfloat16) / shp[1] ** 0.5)
    elif "relative_attention_bias" in k:
        w[k] = torch.randn(shp, device="cuda")
    else:
        w[k] = torch.ones(shp, device="cuda", dtype=torch.bfloat16)
enc = ltxhip.T5TextEncoder(ltxhip.T5EncoderConfig(), w, torch.bfloat16)
del w; torch.cuda.empty_cache()
ids = torch.randint(2, 32000, (1, 128))
out = enc.forward(ids); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = enc.forward(ids)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
params = sum(int(torch.tensor(s).prod()) for s in O.t5_weight_shapes(cfg).values())
print(json.dumps({"op": "t5_xxl_encode", "tokens": 128, "ms": round(ms, 2), "params_B": round(params / 1e9, 2),
                  "weight_stream_GBps": round((params - 32128 * 4096) * 2 / ms / 1e6, 1), "finite": bool(torch.isfinite(out.float()).all())}))
