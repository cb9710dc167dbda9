#!/usr/bin/env python3
"""Whole path at the headline size (512x768x97, 7 distilled steps, synthetic weights): bf16 production mode against the
f32 parity mode of the same engine, PSNR on the [0,255] video (the reference's own pipeline bar is PSNR > 35 dB,
tests/verify_pipeline_parity.rs:7, 48-55)."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import torch, ltxhip
import ltx_oracle as O
from bench import synth_on_device, DISTILLED_SIGMAS
def run():
    dev = "cuda:0"
    F, H, W = 13, 16, 24
    lat = ltxhip.pack_latents(ltxhip.pcg32_randn(42, (1, 128, F, H, W))).to(dev)
    g = torch.Generator().manual_seed(42)
    pe = torch.randn(1, 128, 4096, generator=g).to(dev); pm = torch.zeros(1, 128); pm[:, :32] = 1; pm = pm.to(dev)
    noise = torch.randn(1, 128, F, H, W, generator=torch.Generator().manual_seed(44)).to(dev)
    call = ltxhip.PipelineCall(height=512, width=768, num_frames=97, num_inference_steps=7, sigmas=DISTILLED_SIGMAS, skip_block_list=[], postprocess=True)
    dw = synth_on_device(O.dit_weight_shapes(O.DitConfig()), dev, 1)
    vw = {"decoder." + k: v for k, v in synth_on_device(O.vae_decoder_weight_shapes(O.VaeConfig()), dev, 100).items()}
    res = {}
    for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        dit = ltxhip.LtxVideoTransformer3DModel(ltxhip.LtxVideoTransformer3DModelConfig(), dw, dt)
        vae = ltxhip.AutoencoderKLLtxVideo(ltxhip.AutoencoderKLLtxVideoConfig(), vw, dt)
        pipe = ltxhip.LtxPipeline(dit, vae)
        t0 = time.time()
        if name == "bf16":
            l, v = pipe.call(call, lat, pe, pm, decode_noise=noise)
        else:
            # f32 arithmetic on the SAME inputs the bf16 model sees: the reference rounds the timestep to the model dtype
            # (ltx_transformer.rs:1051: 979 -> 980, 918 -> 920 in bf16), which is an input difference, not an arithmetic one
            sched = ltxhip.FlowMatchEulerDiscreteScheduler(1.0, 0.1)
            ts = sched.set_timesteps(DISTILLED_SIGMAS, None)
            coords = ltxhip.build_video_coords(F, H, W)[None].to(dev)
            l = lat.clone()
            for t in ts:
                tr = float(torch.tensor(float(t)).bfloat16())
                pred = dit.forward(l, pe, [tr], pm, F, H, W, video_coords=coords)
                l = sched.step(pred, t, l)
            v = vae.decode_tokens(l, F, H, W, [0.05], noise, [0.025], postprocess=True)
        torch.cuda.synchronize()
        res[name] = (l.float().cpu(), v.float().cpu(), time.time() - t0)
        del dit, vae, pipe; torch.cuda.empty_cache()
    lb, vb, tb = res["bf16"]; lf, vf, tf = res["f32"]
    mse = float(((vb - vf) ** 2).mean())
    psnr = 100.0 if mse < 1e-10 else 10 * math.log10(255.0 ** 2 / mse)
    lat_rel = float((lb - lf).norm() / lf.norm())
    return {"video_psnr_db_bf16_vs_f32": round(psnr, 2), "latent_rel_l2": round(lat_rel, 5), "video_mean": round(float(vf.mean()), 2),
            "video_std": round(float(vf.std()), 2), "seconds": {"bf16": round(tb, 2), "f32": round(tf, 2)}}


if __name__ == "__main__":
    print(json.dumps(run()))
