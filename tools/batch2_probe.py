#!/usr/bin/env python3
"""Experiment (round 5): C2 as ONE pipeline call of two videos (B = 2) against two calls of one (the headline's form).  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, ROOT)
import torch
import ltxhip
from ltxhip import schema
from bench import synth_on_device


def main():
    dev = "cuda:0"
    pre = ltxhip.get_config_by_version("0.9.8-2b-distilled")
    F, H, W = 13, 16, 24
    call = pre.pipeline_call(512, 768, 97, postprocess=True)
    dit = ltxhip.LtxVideoTransformer3DModel(pre.transformer, synth_on_device(schema.dit_weight_shapes(pre.transformer), dev, 1), torch.bfloat16, 0)
    vae = ltxhip.AutoencoderKLLtxVideo(pre.vae, {"decoder." + k: v for k, v in synth_on_device(schema.vae_decoder_weight_shapes(pre.vae), dev, 100).items()}, torch.bfloat16, 0)
    pipe = ltxhip.LtxPipeline(dit, vae)
    res = {}
    for B in (1, 2):
        lat = ltxhip.pack_latents(ltxhip.pcg32_randn(42, (B, 128, F, H, W))).to(dev)
        pe = torch.randn(B, 128, 4096, generator=torch.Generator().manual_seed(42)).to(dev)
        pm = torch.zeros(B, 128); pm[:, :32] = 1; pm = pm.to(dev)
        noise = torch.randn(B, 128, F, H, W, generator=torch.Generator().manual_seed(44)).to(dev)
        ltxhip.warmup(dit, vae, B, F, H, W, 128)
        for _ in range(2): pipe.call(call, lat, pe, pm, None, None, decode_noise=noise)
        torch.cuda.synchronize()
        for rep in range(2):
            n = 8 // B
            t0 = time.perf_counter()
            for _ in range(n): pipe.call(call, lat, pe, pm, None, None, decode_noise=noise)
            torch.cuda.synchronize()
            res[f"B{B}_frames_per_s_{rep}"] = round(n * B * 97 / (time.perf_counter() - t0), 2)
    print(json.dumps({"what": "C2, eight videos: one per pipeline call vs two per call (B = 2)", **res}))


if __name__ == "__main__":
    main()
