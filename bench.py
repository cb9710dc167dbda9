#!/usr/bin/env python3
"""bench.py — end-to-end LTX-Video-2B hot path on MI355X (BASELINE.json metric).

A "step" = one full pass of the hot path over one batch of synthetic input: the distilled
0.9.8-2B preset's 7 denoise steps (one DiT forward + Euler step each) + denormalize/noise-mix +
3D-VAE decode + postprocess, producing one 512x768x97 video (configs[1] of BASELINE.json).
`value` = frames/sec over the whole job; inputs (latents, embeddings, weights) are resident in
HBM before the timed region.  Multi-GPU: the distilled preset has one forward per step, so the
denoise loop does not shard ("replicas only") — every rank generates its own video,
no data-path collective, scaling = weak.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2] [--no-cpu-baseline]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "candle-video_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[1]: LTX-Video-0.9.8-2B-distilled bf16, 512x768, 97 frames
    "c2": dict(height=512, width=768, num_frames=97, name="LTX-Video-0.9.8-2B-distilled 512x768x97, 7 steps (configs.rs:223-240), untiled VAE decode"),
    # BASELINE.json configs[0]: 256x384, 25 frames (the reference's CPU-runnable case)
    "c1": dict(height=256, width=384, num_frames=25, name="LTX-Video-0.9.8-2B-distilled 256x384x25, 7 steps"),
}
DISTILLED_SIGMAS = [1.0, 0.9937, 0.9875, 0.9812, 0.9750, 0.9094, 0.7250]      # configs.rs:232
PEAK_BF16_TFLOPS = 2500.0    # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def dit_flops(S, D=2048, L=28, K=128, Cin=128, Cout=128, Dc=4096):
    """SURVEY §8d algorithmic FLOPs of one DiT forward."""
    per_layer = 8 * S * D * D + 4 * S * S * D + 4 * S * D * D + 4 * K * D * D + 4 * S * K * D + 16 * S * D * D
    return L * per_layer + 2 * S * Cin * D + 2 * S * D * Cout + 2 * K * Dc * D + 2 * K * D * D + (2 * 256 * D + 2 * D * D + 12 * D * D)


def vae_flops(F, H, W):
    """Σ 54·Cin·Cout·T·H·W over the decoder's 45 convs (SURVEY §8a stage table)."""
    tot = 54 * 128 * 1024 * F * H * W + 10 * 54 * 1024 * 1024 * F * H * W
    t, h, w, cin = F, H, W, 1024
    for ch in (512, 256, 128):
        tot += 54 * cin * (8 * ch) * t * h * w
        t, h, w = 2 * t - 1, 2 * h, 2 * w
        tot += 10 * 54 * ch * ch * t * h * w
        cin = ch
    return tot + 54 * 128 * 48 * t * h * w


def synth_on_device(shapes, dev, seed):
    """Random-init weights of the real architecture directly in HBM (no checkpoints offline); same
    scaling rules as oracle.synth_weights so activations stay O(1)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    out = {}
    for name, shp in shapes.items():
        if name.endswith("timestep_scale_multiplier"):
            w = torch.tensor(1000.0, device=dev)
        elif "norm_q" in name or "norm_k" in name:
            w = 1.0 + 0.1 * torch.randn(shp, generator=g, device=dev)
        elif name.endswith("scale_shift_table"):
            w = torch.randn(shp, generator=g, device=dev) / math.sqrt(shp[-1])
        elif name.endswith(".bias"):
            w = 0.02 * torch.randn(shp, generator=g, device=dev)
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            w = torch.randn(shp, generator=g, device=dev, dtype=torch.bfloat16) / math.sqrt(fan_in)
        out[name] = w
    return out


def cpu_baseline(cfg):
    """The oracle (a port of the reference's CPU path: f32, un-fused, materialised attention scores,
    conv3d as per-frame sums of conv2d) timed on this box's host cores on a bounded sample of the SAME
    workload (10-20 s of CPU work), scaled to frames/sec: DiT forwards with 1 and 3 of the 28 layers at the full token
    count (fixed part + 28 x per-layer time, x7 steps) and a VAE decode of a latent crop (scaled by conv FLOPs)."""
    import ltx_oracle as O
    # 16 threads: torch's small-conv2d / bmm paths get SLOWER with hundreds of threads (measured: the VAE crop
    # took 488 s with 256 threads vs ~1 s with 8); `cores` reports what was actually used.
    ncores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    F, H, W = (cfg["num_frames"] - 1) // 8 + 1, cfg["height"] // 32, cfg["width"] // 32
    S = F * H * W
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, S, 128, generator=g); enc = torch.randn(1, 128, 4096, generator=g)
    mask = torch.zeros(1, 128); mask[:, :32] = 1
    coords = O.build_video_coords(1, F, H, W)
    # DiT forward = fixed part (projections, embeddings, RoPE) + 28 x layer: time a 1-layer and a 3-layer model
    t_n = {}
    for nl in (0, 1, 3):                       # 0 = untimed warm-up of the thread pool / allocator with the 1-layer model
        dcfg = O.DitConfig(num_layers=max(nl, 1))
        dw = O.synth_weights(O.dit_weight_shapes(dcfg), seed=1)
        t0 = time.time()
        O.dit_forward(dw, dcfg, x, enc, torch.tensor([1000.0]), mask, F, H, W, None, coords)
        if nl:
            t_n[nl] = time.time() - t0
        del dw
    t_layer = max((t_n[3] - t_n[1]) / 2.0, 1e-6)
    t_fixed = max(t_n[1] - t_layer, 0.0)
    t_fwd = t_fixed + 28 * t_layer
    vcfg = O.VaeConfig()
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=2)
    cf, chh, cww = 2, 3, 4
    z = torch.randn(1, 128, cf, chh, cww, generator=g)
    t0 = time.time()
    O.decoder_forward(vw, vcfg, z, torch.tensor([0.05]))
    t_crop = time.time() - t0
    t_vae = t_crop * vae_flops(F, H, W) / vae_flops(cf, chh, cww)
    total = 7 * t_fwd + t_vae
    return {"value": cfg["num_frames"] / total, "unit": "frames/sec", "cores": ncores, "kind": "port",
            "sample": f"oracle f32 on host ({t_n[1] + t_n[3] + t_crop:.1f}s of CPU work): DiT forwards with 1 and 3 of 28 layers at S={S} "
                      f"({t_n[1]:.2f}s, {t_n[3]:.2f}s -> {t_fixed:.2f}s + 28 x {t_layer:.2f}s per forward, x7 steps) + VAE decode of a "
                      f"{cf}x{chh}x{cww} latent crop ({t_crop:.2f}s, scaled by conv FLOPs to {F}x{H}x{W}); estimated {total:.1f}s per video"}


def rank_plan(world, rank):
    """Replicas-only sharding (one forward per step in the distilled preset, so the denoise
    loop does not partition): every rank owns an independent video; seeds differ per rank."""
    return {"latent_seed": 42 + rank, "dit_weight_seed": 1 + rank, "vae_weight_seed": 100 + rank, "videos_per_step": 1,
            "total_videos_per_step": world}


def reduce_elapsed(elapsed, dist, device):
    """max over ranks of the timed region (contract: barrier + sync on both sides, MAX over ranks)."""
    if dist is None:
        return elapsed
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_fps(world, steps, frames, elapsed):
    return world * steps * frames / elapsed


def pmc_traffic_bytes(cls):
    """Per-launch memory-side traffic of a kernel class from the newest committed rocprofv3 --pmc summary under profiles/
    (separate FETCH_SIZE / WRITE_SIZE passes, KiB units, FETCH doubled on gfx950: MI355X_MICROARCH.md HBM section).
    bench.py cannot run the profiler on itself; None when no summary is present."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_kernel_stats.json"))):
        try:
            d = json.load(open(f))
            fe, wr = d["pmc"]["FETCH_SIZE"][cls], d["pmc"]["WRITE_SIZE"][cls]
            key = "per_launch_MB_raw" if "per_launch_MB_raw" in fe else "per_launch_MB"
            best = ((2.0 * fe[key] + wr[key]) * 1e6, os.path.basename(f))
        except Exception:
            continue
    return best if best else (None, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-kernel event timing pass")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    import ltxhip
    import ltx_oracle as O

    cfg = CONFIGS[a.config]
    F, H, W = (cfg["num_frames"] - 1) // 8 + 1, cfg["height"] // 32, cfg["width"] // 32
    S = F * H * W
    dcfg = O.DitConfig()
    plan = rank_plan(world, rank)
    dit = ltxhip.LtxVideoTransformer3DModel(ltxhip.LtxVideoTransformer3DModelConfig(), synth_on_device(O.dit_weight_shapes(dcfg), dev, plan["dit_weight_seed"]), torch.bfloat16, local)
    vw = {"decoder." + k: v for k, v in synth_on_device(O.vae_decoder_weight_shapes(O.VaeConfig()), dev, plan["vae_weight_seed"]).items()}
    vae = ltxhip.AutoencoderKLLtxVideo(ltxhip.AutoencoderKLLtxVideoConfig(), vw, torch.bfloat16, local)
    del vw
    torch.cuda.empty_cache()
    pipe = ltxhip.LtxPipeline(dit, vae)
    # synthetic inputs per BASELINE.md §3: PCG32 latents seed 42, embeddings N(0,1) seed 42, mask 32 ones, noise seed 44
    lat = ltxhip.pack_latents(ltxhip.pcg32_randn(plan["latent_seed"], (1, 128, F, H, W))).to(dev)
    g = torch.Generator().manual_seed(42)
    pe = torch.randn(1, 128, 4096, generator=g).to(dev)
    pm = torch.zeros(1, 128); pm[:, :32] = 1; pm = pm.to(dev)
    noise = torch.randn(1, 128, F, H, W, generator=torch.Generator().manual_seed(44)).to(dev)
    call = ltxhip.PipelineCall(height=cfg["height"], width=cfg["width"], num_frames=cfg["num_frames"], num_inference_steps=7,
                               sigmas=DISTILLED_SIGMAS, guidance_scale=1.0, stg_scale=0.0, skip_block_list=[], postprocess=True)

    def step():
        return pipe.call(call, lat, pe, pm, decode_noise=noise)

    # Initialisation, not a warm-up step: the first call of every GEMM shape measures the candidate plans (tile / kernel)
    # and caches the winner, and the models size their workspaces.  Done once here so that the W warm-up steps and the K
    # timed steps below all run the steady-state path even when the driver passes --warmup 0.
    step()
    for _ in range(a.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    dit_ms = vae_ms = 0.0
    for _ in range(a.steps):
        _, video = step()
        dit_ms += pipe.last_timing_ms[0]; vae_ms += pipe.last_timing_ms[2]
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = reduce_elapsed(elapsed, dist, dev)
    assert torch.isfinite(video).all()

    out = None
    if rank == 0:
        fps = job_fps(world, a.steps, cfg["num_frames"], elapsed)
        fl_dit, fl_vae = dit_flops(S), vae_flops(F, H, W)
        out = {"metric": "frames/sec end-to-end LTX-Video-2B 512x768x97; DiT step ms; VAE decode ms", "value": fps, "unit": "frames/sec",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * elapsed / a.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": cfg["name"], "latent_grid": [F, H, W], "tokens": S, "text_tokens": 128,
                          "denoise_steps": 7, "parallelism": f"replicas x{world} (no data-path collective)"},
               "dit_step_ms": dit_ms / (7 * a.steps), "vae_decode_ms": vae_ms / a.steps,
               "dit_tflops": fl_dit / (dit_ms / (7 * a.steps) * 1e-3) / 1e12, "vae_tflops": fl_vae / (vae_ms / a.steps * 1e-3) / 1e12,
               "algorithmic_tflop_per_video": (7 * fl_dit + fl_vae) / 1e12}
        if not a.no_prof:
            # separate, untimed pass with hipEvents around every launch of the heavy kernel classes (on their stream)
            ltxhip.prof_enable(True)
            step()
            kinds = {"gemm_big_kernel/gemm_p8_kernel<bf16> (Linear GEMMs, tile per shape)": 0,
                     "gemm_big_kernel/gemm_p8_kernel<bf16,conv> (conv3d implicit GEMM)": 1, "attn_pipe64_kernel (self attention)": 2,
                     "attn_bf16_kernel<64> (cross attention)": 3, "rownorm_kernel<bf16>": 4}
            per = {}
            for name, k in kinds.items():
                ms, work, cnt = ltxhip.prof_report(k)
                per[name] = {"ms_total": ms, "launches": cnt, "avg_ms": ms / max(cnt, 1),
                             ("GB/s" if k == 4 else "TFLOP/s"): (work / 1e9 if k == 4 else work / 1e12) / max(ms * 1e-3, 1e-12)}
            ltxhip.prof_enable(False)
            dom = max((n for n in per if "rownorm" not in n), key=lambda n: per[n]["ms_total"])
            ach = per[dom]["TFLOP/s"]
            traffic, tsrc = pmc_traffic_bytes("conv3d implicit GEMM" if "conv" in dom else "linear GEMM")
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes per launch (L2-miss side: HBM + Infinity Cache)",
                               "traffic_source": tsrc, "avg_launch_ms": per[dom]["avg_ms"],
                               "launches_per_video": per[dom]["launches"]}
            out["kernels"] = per
            Fh, Hh, Wh = F, H, W
            out["gemm_plans"] = {"qkv": ltxhip.ops.gemm_plan(S, 6144, 2048), "attn_out/q2/out2": ltxhip.ops.gemm_plan(S, 2048, 2048),
                                 "ff1": ltxhip.ops.gemm_plan(S, 8192, 2048), "ff2": ltxhip.ops.gemm_plan(S, 2048, 8192),
                                 "vae_mid_1024": ltxhip.ops.gemm_plan(Fh * Hh * Wh, 1024, 1024, 1, 27, Fh, Hh, Wh)}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
