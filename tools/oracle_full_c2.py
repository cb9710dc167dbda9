#!/usr/bin/env python3
"""One FULL run of the CPU oracle (oracle/ltx_oracle.py, torch-CPU f32: a port of the reference's CPU path, not its binary) on the
headline workload - LTX-Video-0.9.8-2B-distilled 512x768x97, 7 steps + untiled VAE decode, BASELINE.json configs[1] - on the host
cores of the machine it runs on.  Writes one JSON record (seconds, frames/s, CPU model, threads); bench.py's cpu_baseline cites
the committed copy (profiles/r5_oracle_c2_on_gpu_box.json) as `full_c2_run` beside its bounded-sample ESTIMATE.
    python3 tools/oracle_full_c2.py [out.json] [threads]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ltx_oracle as O  # noqa: E402
from bench import DISTILLED_SIGMAS, dit_flops, vae_flops, host_machine  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "oracle_c2_full.json")
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    dcfg, vcfg = O.DitConfig(), O.VaeConfig()
    dw = O.synth_weights(O.dit_weight_shapes(dcfg), seed=31)
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=32)
    F, H, W = 13, 16, 24
    lat = O.pack_latents(O.Pcg32(42, 1442695040888963407).randn((1, 128, F, H, W)))
    pe = torch.randn(1, 128, 4096, generator=torch.Generator().manual_seed(42)); pm = torch.zeros(1, 128); pm[:, :32] = 1
    noise = torch.randn(1, 128, F, H, W, generator=torch.Generator().manual_seed(44))
    args = O.PipelineArgs(height=512, width=768, num_frames=97, num_inference_steps=7, sigmas=DISTILLED_SIGMAS, guidance_scale=1.0, stg_scale=0.0,
                          decode_timestep=0.05, decode_noise_scale=0.025)
    O.dit_forward(dw, dcfg, lat[:, :8], pe[:, :8], torch.tensor([1000.0]), pm[:, :8], 1, 2, 4, None, O.build_video_coords(1, 1, 2, 4))   # thread pool warm-up
    t0 = time.time()
    video = O.pipeline_call(dw, dcfg, vw, vcfg, torch.zeros(128), torch.ones(128), args, lat, pe, pm, None, None, noise, torch.float32)
    sec = time.time() - t0
    assert torch.isfinite(video).all() and tuple(video.shape) == (1, 3, 97, 512, 768), video.shape
    fl = 7 * dit_flops(F * H * W) + vae_flops(F, H, W)
    rec = {"what": "oracle/ltx_oracle.py pipeline_call, f32, BASELINE configs[1] in full (7 distilled steps + untiled decode), measured, not scaled",
           "seconds": sec, "frames_per_sec": 97.0 / sec, "tflop": fl / 1e12, "cpu_tflops": fl / sec / 1e12, "torch_threads": threads,
           "machine": host_machine(), "torch": torch.__version__}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
