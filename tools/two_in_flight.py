#!/usr/bin/env python3
"""Experiment (round 5): two videos in flight on two HIP streams, one host thread each, against the same two videos one after
the other on one stream.  Motive: a one-round grid leaves the chip partly idle at every kernel boundary (ramp + tail of ~2444
launches per video); a second, independent video could fill those CUs.  Prints one JSON line.
    python3 tools/two_in_flight.py [videos_per_arm]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, ROOT)
import torch
import ltxhip
from ltxhip import schema
from bench import synth_on_device


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = "cuda:0"
    pre = ltxhip.get_config_by_version("0.9.8-2b-distilled")
    F, H, W = 13, 16, 24
    call = pre.pipeline_call(512, 768, 97, postprocess=True)
    pipes, inputs = [], []
    for r in range(2):
        dit = ltxhip.LtxVideoTransformer3DModel(pre.transformer, synth_on_device(schema.dit_weight_shapes(pre.transformer), dev, 1 + r), torch.bfloat16, 0)
        vae = ltxhip.AutoencoderKLLtxVideo(pre.vae, {"decoder." + k: v for k, v in synth_on_device(schema.vae_decoder_weight_shapes(pre.vae), dev, 100 + r).items()}, torch.bfloat16, 0)
        pipes.append(ltxhip.LtxPipeline(dit, vae))
        lat = ltxhip.pack_latents(ltxhip.pcg32_randn(42 + r, (1, 128, F, H, W))).to(dev)
        pe = torch.randn(1, 128, 4096, generator=torch.Generator().manual_seed(42)).to(dev)
        pm = torch.zeros(1, 128); pm[:, :32] = 1; pm = pm.to(dev)
        noise = torch.randn(1, 128, F, H, W, generator=torch.Generator().manual_seed(44)).to(dev)
        inputs.append((lat, pe, pm, noise))
        ltxhip.warmup(dit, vae, 1, F, H, W, 128)
    def run(r, count, stream):
        with torch.cuda.stream(stream):
            lat, pe, pm, noise = inputs[r]
            for _ in range(count): pipes[r].call(call, lat, pe, pm, None, None, decode_noise=noise)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    run(0, 1, s0); run(1, 1, s1); torch.cuda.synchronize()
    res = {}
    for rep in range(2):
        t0 = time.perf_counter(); run(0, n, s0); run(1, n, s0); torch.cuda.synchronize(); res[f"sequential_{rep}"] = 2 * n * 97 / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(r, n, s)) for r, s in ((0, s0), (1, s1))]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); res[f"two_in_flight_{rep}"] = 2 * n * 97 / (time.perf_counter() - t0)
    print(json.dumps({"what": "frames/s, C2, two videos one after the other on one stream vs two in flight on two streams (two host threads)", "videos_per_arm": 2 * n, **{k: round(v, 2) for k, v in res.items()}}))


if __name__ == "__main__":
    main()
