#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Run IN THE BUILD CONTAINER (needs /root/reference for the two reference-script pins):

    python tools/gen_fixtures.py

Two kinds of vectors:
  * ref_*.safetensors    — outputs of the reference's own runnable torch-only scripts
                           (scripts/gen_guidance_ref.py, scripts/gen_latent_norm_ref.py), executed
                           from /root/reference in a temp dir; these PIN the oracle.
  * oracle_*.safetensors — inputs + weights + outputs of oracle/ltx_oracle.py (f32 and bf16
                           modes) for tiny DiT / VAE / pipeline configs modelled on the
                           reference's tests (tests/verify_dit_parity.rs:24-39,
                           tests/verify_rope_parity.rs:537-552, tests/vae_tests.rs:119-180);
                           these pin the oracle against drift and are what the GPU path is
                           compared with on a box where /root/reference does not exist.
Nothing here copies reference source; fixtures are data only.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch
from safetensors.torch import load_file, save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ltx_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def c(t):
    return t.detach().contiguous().clone()


def weights_checksum(w):
    """Order-independent fingerprint so a test can tell 'RNG stream changed' from 'kernel is wrong'."""
    return torch.tensor([sum(float(v.double().sum()) for v in w.values()), sum(float(v.double().abs().sum()) for v in w.values())],
                        dtype=torch.float64)


def ref_scripts():
    """Run the reference's torch-only generators and keep (slices of) their outputs."""
    with tempfile.TemporaryDirectory() as td:
        for s in ("gen_guidance_ref.py", "gen_latent_norm_ref.py"):
            subprocess.run([sys.executable, os.path.join(REF, "scripts", s)], cwd=td, check=True, stdout=subprocess.DEVNULL)
        g = load_file(os.path.join(td, "gen_guidance_ref.safetensors"))
        save_file({k: c(v) for k, v in g.items()}, os.path.join(GOLD, "ref_guidance.safetensors"),
                  metadata={"source": "reference scripts/gen_guidance_ref.py (seed 42), executed unmodified"})
        n = load_file(os.path.join(td, "gen_latent_norm_ref.safetensors"))
        sl = (slice(None), slice(None), slice(0, 3), slice(0, 4), slice(0, 6))
        keep = {"latents": c(n["latents"][sl]), "normalized": c(n["normalized"][sl]), "denormalized": c(n["denormalized"][sl]),
                "latents_mean": c(n["latents_mean"]), "latents_std": c(n["latents_std"]), "scaling_factor": c(n["scaling_factor"])}
        save_file(keep, os.path.join(GOLD, "ref_latent_norm.safetensors"),
                  metadata={"source": "reference scripts/gen_latent_norm_ref.py (seed 42), executed unmodified; tensors sliced [:, :, :3, :4, :6]"})


def ref_scripts_imported():
    """Three more of the reference's torch-only scripts, executed UNMODIFIED from /root/reference (runpy, in a temp cwd) and
    asked for outputs of their own functions:
      scripts/test_unpatchify.py   - the decoder's permute(0,1,5,2,6,4,7,3) on its index-coded tensor (module globals x, out)
      scripts/verify_rng.py        - its Pcg32 class: u32 stream and randn for the seed / increment of main.rs:568
      scripts/test_rope_rotation.py - rust_apply_rotary_emb_linear ("what Rust apply_rotary_emb_linear should do")"""
    import contextlib, io, runpy
    with tempfile.TemporaryDirectory() as td:
        cwd = os.getcwd(); os.chdir(td)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                up = runpy.run_path(os.path.join(REF, "scripts", "test_unpatchify.py"), run_name="ref_unpatchify")
                rng = runpy.run_path(os.path.join(REF, "scripts", "verify_rng.py"), run_name="ref_rng")
                rope = runpy.run_path(os.path.join(REF, "scripts", "test_rope_rotation.py"), run_name="ref_rope")
        finally:
            os.chdir(cwd)
    save_file({"x": c(up["x"]), "out": c(up["out"])}, os.path.join(GOLD, "ref_unpatchify.safetensors"),
              metadata={"source": "reference scripts/test_unpatchify.py executed unmodified: its x [1,48,2,4,6] and python_unpatchify(x, 4, 1)"})
    P = rng["Pcg32"]
    r = P(42, 1442695040888963407)
    u32 = torch.tensor([r.next_u32() for _ in range(64)], dtype=torch.int64)
    r2 = P(42, 54)                                          # the PCG paper's demo seed / sequence
    u32_demo = torch.tensor([r2.next_u32() for _ in range(6)], dtype=torch.int64)
    randn = P(42, 1442695040888963407).randn((257,))       # odd count: the dropped second Box-Muller value
    save_file({"u32": u32, "u32_seed42_seq54": u32_demo, "randn": c(randn.float())}, os.path.join(GOLD, "ref_rng.safetensors"),
              metadata={"source": "reference scripts/verify_rng.py executed unmodified: its Pcg32(42, 1442695040888963407) u32 stream / randn((257,)); Pcg32(42, 54) first six"})
    g = torch.Generator().manual_seed(9)
    # tables as the model builds them: one (cos, sin) per channel PAIR, repeat_interleave 2 (ltx_transformer.rs:436-524)
    x = torch.randn(1, 24, 96, generator=g)
    cs = (torch.rand(1, 24, 48, generator=g) * 2 - 1).repeat_interleave(2, -1); sn = (torch.rand(1, 24, 48, generator=g) * 2 - 1).repeat_interleave(2, -1)
    save_file({"x": x, "cos": cs, "sin": sn, "out": c(rope["rust_apply_rotary_emb_linear"](x, (cs, sn))),
               "out_diffusers": c(rope["diffusers_apply_rotary_emb"](x, (cs, sn)))}, os.path.join(GOLD, "ref_rope_rotation.safetensors"),
              metadata={"source": "reference scripts/test_rope_rotation.py executed unmodified: rust_apply_rotary_emb_linear / diffusers_apply_rotary_emb on seeded inputs"})


def ref_rope_tables():
    """The RoPE TABLE and GRID pinned to the reference's own torch-only scripts, executed UNMODIFIED
    from /root/reference with runpy and asked for the outputs of their own functions:
      scripts/compare_rope_freqs.py - rust_compute_freqs ("how Rust computes RoPE frequencies") and diffusers_compute_freqs:
                                      frequency layout, transpose-then-flatten, repeat_interleave 2, LEFT pad of dim % 6
      scripts/compare_rope_grid.py  - rust_expected_grid / diffusers_rope_grid: (f, h, w) order, f slowest
      scripts/debug_rope.py         - prepare_video_coords_debug (the grid with rope_interpolation_scale * patch / base,
                                      ltx_transformer.rs:373-433) and compute_freqs_debug
      scripts/test_rng.py           - its top-level Pcg32(42, 1442695040888963407) Box-Muller values (both z0 and z1 kept)
    Full-width tables for dim 2048 (pad 2) and 4096 (pad 4); rows kept small (24 + 12)."""
    import contextlib, io, runpy
    with tempfile.TemporaryDirectory() as td:
        cwd = os.getcwd(); os.chdir(td)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                fr = runpy.run_path(os.path.join(REF, "scripts", "compare_rope_freqs.py"), run_name="ref_rope_freqs")
                gr = runpy.run_path(os.path.join(REF, "scripts", "compare_rope_grid.py"), run_name="ref_rope_grid")
                db = runpy.run_path(os.path.join(REF, "scripts", "debug_rope.py"), run_name="ref_debug_rope")
                rg = runpy.run_path(os.path.join(REF, "scripts", "test_rng.py"), run_name="ref_test_rng")
                F, H, W = 2, 3, 4
                scale = (8.0 / 25.0, 32.0, 32.0)                     # debug_rope.py main(): the pipeline's interpolation scale
                grid_scaled = db["prepare_video_coords_debug"](2, F, H, W, scale)          # [2, 24, 3], base (20, 2048, 2048)
                grid_raw = gr["diffusers_rope_grid"](1, F, H, W)                            # raw (f, h, w) indices
                grid_frac = gr["rust_expected_grid"](1, F, H, W)                            # index / (extent - 1)
                g = torch.Generator().manual_seed(31)
                grid_rand = torch.rand(1, 12, 3, generator=g) * 1.5                         # fractional positions past 1 too
                out = {"grid_scaled": c(grid_scaled), "grid_raw": c(grid_raw), "grid_frac": c(grid_frac), "grid_rand": c(grid_rand),
                       "scale": torch.tensor(scale), "fhw": torch.tensor([F, H, W])}
                for dim in (2048, 4096):
                    for gname, grid in (("scaled", grid_scaled[:1]), ("rand", grid_rand)):
                        cr, sr = fr["rust_compute_freqs"](dim, 10000.0, grid)
                        cd, sd = fr["diffusers_compute_freqs"](dim, 10000.0, grid)
                        out[f"rust_cos_{dim}_{gname}"], out[f"rust_sin_{dim}_{gname}"] = c(cr), c(sr)
                        # the diffusers form of the same table: kept half-width (its channel pairs are equal, asserted here)
                        assert torch.equal(cd[..., 0::2], cd[..., 1::2]) and torch.equal(sd[..., 0::2], sd[..., 1::2])
                        out[f"diffusers_cos_{dim}_{gname}_even"], out[f"diffusers_sin_{dim}_{gname}_even"] = c(cd[..., 0::2]), c(sd[..., 0::2])
                cdbg, sdbg = db["compute_freqs_debug"](grid_scaled[:1], dim=2048)
                out["debug_cos_2048_scaled_even"], out["debug_sin_2048_scaled_even"] = c(cdbg[..., 0::2]), c(sdbg[..., 0::2])
                out["test_rng_values"] = torch.tensor(rg["values"][:10], dtype=torch.float64)
        finally:
            os.chdir(cwd)
    save_file(out, os.path.join(GOLD, "ref_rope_table.safetensors"),
              metadata={"source": "reference scripts/compare_rope_freqs.py, compare_rope_grid.py, debug_rope.py, test_rng.py executed unmodified "
                                  "(runpy): rust_compute_freqs / diffusers_compute_freqs(dim, 10000, grid) for dim 2048 and 4096 on "
                                  "prepare_video_coords_debug(2, 2, 3, 4, (0.32, 32, 32)) and on a seeded random grid; rust_expected_grid / "
                                  "diffusers_rope_grid(1, 2, 3, 4); test_rng.py's first ten Box-Muller values"})


DIT_CASES = {
    # tests/verify_dit_parity.rs:24-39 config (2 layers, 2 heads x 16, dims 32), smaller grid, no mask, rope scale (1,1,1)
    "A": dict(cfg=dict(in_channels=32, out_channels=32, num_attention_heads=2, attention_head_dim=16, cross_attention_dim=32,
                       num_layers=2, caption_channels=32), B=1, grid=(8, 16, 16), K=10, mask=None, rope_scale=(1.0, 1.0, 1.0),
              coords=False, t=500.0),
    # tests/verify_rope_parity.rs:537-552 config (2 layers, 4 heads x 16), grid (4,8,8), K=16, partially masked, video_coords
    "B": dict(cfg=dict(in_channels=32, out_channels=32, num_attention_heads=4, attention_head_dim=16, cross_attention_dim=64,
                       num_layers=2, caption_channels=32), B=2, grid=(4, 8, 8), K=16, mask=[16, 5], rope_scale=None,
              coords=True, t=979.0, skip_layer_mask=True),
    # production head_dim 64, C1's latent grid 4x8x12 (S=384), K=128 with 32 valid tokens (BASELINE.md synthetic inputs)
    "C": dict(cfg=dict(in_channels=128, out_channels=128, num_attention_heads=2, attention_head_dim=64, cross_attention_dim=128,
                       num_layers=2, caption_channels=64), B=1, grid=(4, 8, 12), K=128, mask=[32], rope_scale=None,
              coords=True, t=918.0, skip_blocks=[1]),
}


def dit_case(name, spec):
    cfg = O.DitConfig(**spec["cfg"])
    w = O.synth_weights(O.dit_weight_shapes(cfg), seed=ord(name))
    g = torch.Generator().manual_seed(100 + ord(name))
    B, (F, H, W), K = spec["B"], spec["grid"], spec["K"]
    S = F * H * W
    hidden = torch.randn(B, S, cfg.in_channels, generator=g)
    enc = torch.randn(B, K, cfg.caption_channels, generator=g)
    mask = None
    if spec["mask"] is not None:
        mask = torch.zeros(B, K)
        for b, nv in enumerate(spec["mask"]):
            mask[b, :nv] = 1.0
    coords = O.build_video_coords(B, F, H, W) if spec["coords"] else None
    t = torch.full((B,), spec["t"])
    slm = None
    if spec.get("skip_layer_mask"):
        slm = torch.zeros(cfg.num_layers, B)
        slm[0, 1] = 1.0            # layer 0 skipped for batch row 1 only (mixed row -> blend path)
        slm[1, :] = 1.0            # layer 1 skipped for everyone (identity shortcut)
    skip = spec.get("skip_blocks", [])
    out = {"hidden": hidden, "enc": enc, "timestep": t}
    if mask is not None:
        out["mask"] = mask
    if coords is not None:
        out["coords"] = coords
    if slm is not None:
        out["skip_layer_mask"] = slm
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        wd = {k: v.to(dt) for k, v in w.items()}
        y = O.dit_forward(wd, cfg, hidden, enc, t, mask, F, H, W, spec["rope_scale"], coords, slm, skip, dt)
        out["out_" + tag] = y.float()
    for k, v in w.items():
        out["w." + k] = v
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, f"oracle_dit_{name}.safetensors"),
              metadata={"cfg": repr(spec["cfg"]), "grid": repr(spec["grid"]), "rope_scale": repr(spec["rope_scale"]),
                        "skip_blocks": repr(skip), "source": "oracle/ltx_oracle.py dit_forward"})


VAE_CFG = dict(latent_channels=8, decoder_block_out_channels=(32, 64, 128), decoder_layers_per_block=(1, 1, 1, 2))


def vae_case():
    cfg = O.VaeConfig(**VAE_CFG)
    w = O.synth_weights(O.vae_decoder_weight_shapes(cfg), seed=7)
    g = torch.Generator().manual_seed(77)
    z = torch.randn(1, 8, 2, 3, 4, generator=g)
    out = {"z": z, "timestep": torch.tensor([0.05])}
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        wd = {k: v.to(dt) for k, v in w.items()}
        out["out_" + tag] = O.decoder_forward(wd, cfg, z, torch.tensor([0.05]), dt).float()
    # tiled + framewise decode with shrunken tile parameters (latent tile 2x2, stride 1; temporal min 1 / stride 1 latent frame)
    z2 = torch.randn(1, 8, 4, 3, 3, generator=g)
    tcfg = O.VaeConfig(**VAE_CFG, tile_sample_min_height=64, tile_sample_min_width=64, tile_sample_stride_height=32,
                       tile_sample_stride_width=32, tile_sample_min_num_frames=16, tile_sample_stride_num_frames=8)
    out["z_tiled"] = z2
    out["out_tiled_f32"] = O.vae_decode(w, tcfg, z2, torch.tensor([0.05]), torch.float32, use_tiling=True, use_framewise_decoding=True)
    out["out_spatial_tiled_f32"] = O.vae_decode(w, tcfg, z2[:, :, :2], torch.tensor([0.05]), torch.float32, use_tiling=True, use_framewise_decoding=False)
    out["out_tiled_f32"] = out["out_tiled_f32"][..., ::2, ::2]
    out["out_spatial_tiled_f32"] = out["out_spatial_tiled_f32"][..., ::2, ::2]
    out["weights_checksum"] = weights_checksum(w)       # weights are re-derived from seed 7 (O.synth_weights)
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, "oracle_vae.safetensors"),
              metadata={"cfg": repr(VAE_CFG), "source": "oracle/ltx_oracle.py decoder_forward / vae_decode"})


def ops_case():
    g = torch.Generator().manual_seed(5)
    out = {}
    # conv3d known-answer [1,8,5,6,6] -> 16, causal and non-causal
    x = torch.randn(1, 8, 5, 6, 6, generator=g)
    w = torch.randn(16, 8, 3, 3, 3, generator=g) / 14.7
    b = torch.randn(16, generator=g) * 0.1
    out.update({"conv_x": x, "conv_w": w, "conv_b": b,
                "conv_y_noncausal": O.causal_conv3d(x, w, b, False), "conv_y_causal": O.causal_conv3d(x, w, b, True)})
    # upsampler axis-order KAT modelled on tests/vae_tests.rs:119-180: identity-free conv (zero weight) so the
    # output is exactly bias + residual; x value = src_t*100 + packed_channel makes every permutation visible
    cin, cout = 16, 8 * 4     # residual repeats = 32/16 = 2
    xs = torch.zeros(1, cin, 2, 2, 3)
    for t in range(2):
        for ch in range(cin):
            xs[0, ch, t] = t * 100 + ch + 0.01 * torch.arange(6).reshape(2, 3)
    pw = {"conv.conv.weight": torch.zeros(cout, cin, 3, 3, 3), "conv.conv.bias": torch.arange(cout, dtype=torch.float32) * 1000.0}
    out.update({"up_x": xs, "up_bias": pw["conv.conv.bias"], "up_y": O.upsampler(pw, "", xs, cout // 8, False)})
    # RoPE rows for the grids of tests/verify_rope_parity.rs (dim 2048): checksums + a 64-row slice
    for gi, (F, H, W) in enumerate([(2, 8, 8), (13, 16, 24)]):
        coords = O.build_video_coords(1, F, H, W)
        cos, sin = O.rope_cos_sin(2048, 1, F, H, W, None, coords)
        idx = torch.linspace(0, F * H * W - 1, 64).long()
        out[f"rope{gi}_rows"] = idx
        out[f"rope{gi}_cos"] = cos[0, idx]
        out[f"rope{gi}_sin"] = sin[0, idx]
        out[f"rope{gi}_sum"] = torch.stack([cos.double().sum(), sin.double().sum()]).float()
    # PCG32: first u32s and gaussians for seed 42 (main.rs:568 increment)
    r = O.Pcg32(42, 1442695040888963407)
    out["pcg_u32"] = torch.tensor([r.next_u32() for _ in range(16)], dtype=torch.int64)
    out["pcg_randn"] = O.Pcg32(42, 1442695040888963407).randn((32,))
    # scheduler: distilled sigma list (configs.rs:232) and 40-step linspace with mu(S) for the three BASELINE grids
    s = O.FlowMatchEulerScheduler()
    ts = s.set_timesteps(sigmas=[1.0, 0.9937, 0.9875, 0.9812, 0.9750, 0.9094, 0.7250], mu=0.0)
    out["sched_distilled_sigmas"] = torch.from_numpy(s.sigmas.copy())
    out["sched_distilled_timesteps"] = torch.tensor(ts, dtype=torch.int64)
    for S in (384, 4992, 17556):
        lin = list(O.FlowMatchEulerScheduler._linspace(1.0, 1.0 / 40, 40))
        ts = s.set_timesteps(sigmas=lin, mu=O.calculate_shift(S))
        out[f"sched40_S{S}_sigmas"] = torch.from_numpy(s.sigmas.copy())
        out[f"sched40_S{S}_timesteps"] = torch.tensor(ts, dtype=torch.int64)
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, "oracle_ops.safetensors"),
              metadata={"source": "oracle/ltx_oracle.py primitives"})


def pipeline_case():
    """3-step CFG(3.0)+STG(1.0)+rescale(0.7) trajectory on tiny models (the 0.9.5 preset's guidance, configs.rs:163-180)."""
    dcfg = O.DitConfig(in_channels=8, out_channels=8, num_attention_heads=2, attention_head_dim=16, cross_attention_dim=32,
                       num_layers=3, caption_channels=32)
    vcfg = O.VaeConfig(**VAE_CFG)
    dw = O.synth_weights(O.dit_weight_shapes(dcfg), seed=11)
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=12)
    g = torch.Generator().manual_seed(13)
    args = O.PipelineArgs(height=64, width=96, num_frames=9, num_inference_steps=3, guidance_scale=3.0, guidance_rescale=0.7,
                          stg_scale=1.0, skip_block_list=[1], decode_timestep=0.05, decode_noise_scale=0.025)
    F, H, W = 2, 2, 3
    lat = O.pack_latents(O.Pcg32(42, 1442695040888963407).randn((1, 8, F, H, W)))
    pe = torch.randn(1, 16, 32, generator=g); pm = torch.zeros(1, 16); pm[:, :9] = 1
    ne = torch.randn(1, 16, 32, generator=g); nm = torch.zeros(1, 16); nm[:, :4] = 1
    noise = torch.randn(1, 8, F, H, W, generator=g)
    mean = torch.randn(8, generator=g) * 0.1
    std = 1.0 + 0.1 * torch.randn(8, generator=g).abs()
    traj = []
    video = O.pipeline_call(dw, dcfg, vw, vcfg, mean, std, args, lat, pe, pm, ne, nm, noise, torch.float32, trajectory=traj)
    out = {"latents": lat, "prompt_embeds": pe, "prompt_mask": pm, "neg_embeds": ne, "neg_mask": nm, "decode_noise": noise,
           "latents_mean": mean, "latents_std": std, "video": video, "trajectory": torch.stack(traj)}
    out["dit_weights_checksum"] = weights_checksum(dw)  # seeds 11 / 12 (O.synth_weights)
    out["vae_weights_checksum"] = weights_checksum(vw)
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, "oracle_pipeline.safetensors"),
              metadata={"source": "oracle/ltx_oracle.py pipeline_call", "args": repr(args)})


C1 = dict(height=256, width=384, num_frames=25)           # BASELINE.json configs[0]
C1_SIGMAS = [1.0, 0.9937, 0.9875, 0.9812, 0.9750, 0.9094, 0.7250]      # configs.rs:232 (distilled)


def c1_inputs():
    """Synthetic inputs of BASELINE.md section 3 at C1 geometry; shared by the generator and tests/test_gpu_c1.py."""
    F, H, W = 4, 8, 12
    lat = O.pack_latents(O.Pcg32(42, 1442695040888963407).randn((1, 128, F, H, W)))
    pe = torch.randn(1, 128, 4096, generator=torch.Generator().manual_seed(42))
    pm = torch.zeros(1, 128); pm[:, :32] = 1
    noise = torch.randn(1, 128, F, H, W, generator=torch.Generator().manual_seed(44))
    g = torch.Generator().manual_seed(45)
    mean = torch.randn(128, generator=g) * 0.1
    std = 1.0 + 0.1 * torch.randn(128, generator=g).abs()
    return lat, pe, pm, noise, mean, std


def c1_case():
    """BASELINE config C1 in FULL: the 2B DiT (28 layers, D = 2048, 32 x 64 heads) and the full VAE decoder with seeded
    synthetic weights (seeds 31 / 32, re-derived on the GPU box by O.synth_weights), distilled 7 steps + decode at
    256x384x25 (t2v_pipeline.rs:627-1073, configs.rs:223-240), in f32 - and once more in f32 with the model seeing the
    timesteps as a bf16 model does (ltx_transformer.rs:1051), the reference for the production bf16 mode.
    Committed: final latents, the latents after the first step, a strided slice + moments of the video."""
    import time
    dcfg, vcfg = O.DitConfig(), O.VaeConfig()
    dw = O.synth_weights(O.dit_weight_shapes(dcfg), seed=31)
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=32)
    lat, pe, pm, noise, mean, std = c1_inputs()
    args = O.PipelineArgs(height=C1["height"], width=C1["width"], num_frames=C1["num_frames"], num_inference_steps=7, sigmas=C1_SIGMAS,
                          guidance_scale=1.0, stg_scale=0.0, decode_timestep=0.05, decode_noise_scale=0.025)
    out = {"dit_weights_checksum": weights_checksum(dw), "vae_weights_checksum": weights_checksum(vw)}
    for tag, cast in (("f32", None), ("f32_bf16ts", torch.bfloat16)):
        traj = []
        t0 = time.time()
        video = O.pipeline_call(dw, dcfg, vw, vcfg, mean, std, args, lat, pe, pm, None, None, noise, torch.float32, trajectory=traj,
                                timestep_cast=cast)
        dt = time.time() - t0
        out[f"latents_{tag}"] = traj[-1]
        out[f"latents_step1_{tag}"] = traj[0]
        out[f"video_slice_{tag}"] = video[:, :, ::4, ::8, ::8]
        out[f"video_moments_{tag}"] = torch.tensor([float(video.double().mean()), float(video.double().std()), float(video.double().abs().sum())], dtype=torch.float64)
        out[f"oracle_seconds_{tag}"] = torch.tensor([dt], dtype=torch.float64)
        print(f"C1 oracle {tag}: {dt:.1f} s, video mean {float(video.mean()):.2f} std {float(video.std()):.2f}", flush=True)
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, "oracle_c1.safetensors"),
              metadata={"source": "oracle/ltx_oracle.py pipeline_call at BASELINE C1 (full 2B DiT + VAE decoder, synthetic weights seeds 31/32)",
                        "args": repr(args), "video_slice": "[:, :, ::4, ::8, ::8] of the [1,3,25,256,384] post-processed video"})


def c2_case():
    """THE headline config in full: BASELINE C2 = 0.9.8-2B-distilled at 512x768x97 (S = 4992), 7 distilled steps + the
    untiled 48-TFLOP decode, full 2B DiT + VAE (weights of c1_case), synthetic inputs of BASELINE.md section 3 - in f32 and
    once more with the model seeing bf16-rounded timesteps (ltx_transformer.rs:1051).  About 10 minutes of host time per
    run.  Committed: every 8th token of the final latents + their moments, a strided slice + moments of the video."""
    import time
    dcfg, vcfg = O.DitConfig(), O.VaeConfig()
    dw = O.synth_weights(O.dit_weight_shapes(dcfg), seed=31)
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=32)
    F, H, W = 13, 16, 24
    lat = O.pack_latents(O.Pcg32(42, 1442695040888963407).randn((1, 128, F, H, W)))
    _, pe, pm, _, mean, std = c1_inputs()
    noise = torch.randn(1, 128, F, H, W, generator=torch.Generator().manual_seed(44))
    args = O.PipelineArgs(height=512, width=768, num_frames=97, num_inference_steps=7, sigmas=C1_SIGMAS,
                          guidance_scale=1.0, stg_scale=0.0, decode_timestep=0.05, decode_noise_scale=0.025)
    out = {"dit_weights_checksum": weights_checksum(dw), "vae_weights_checksum": weights_checksum(vw)}
    for tag, cast in (("f32", None), ("f32_bf16ts", torch.bfloat16)):
        traj = []
        t0 = time.time()
        video = O.pipeline_call(dw, dcfg, vw, vcfg, mean, std, args, lat, pe, pm, None, None, noise, torch.float32, trajectory=traj, timestep_cast=cast)
        dt = time.time() - t0
        l = traj[-1]
        out[f"latents_sub_{tag}"] = l[:, ::8]
        out[f"latents_moments_{tag}"] = torch.tensor([float(l.double().sum()), float(l.double().abs().sum()), float(l.double().pow(2).sum())], dtype=torch.float64)
        out[f"video_slice_{tag}"] = video[:, :, ::8, ::16, ::16]
        out[f"video_moments_{tag}"] = torch.tensor([float(video.double().mean()), float(video.double().std()), float(video.double().abs().sum())], dtype=torch.float64)
        out[f"oracle_seconds_{tag}"] = torch.tensor([dt], dtype=torch.float64)
        print(f"C2 oracle {tag}: {dt:.1f} s, video mean {float(video.mean()):.2f} std {float(video.std()):.2f}", flush=True)
        del video, traj
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, "oracle_c2.safetensors"),
              metadata={"source": "oracle/ltx_oracle.py pipeline_call at BASELINE C2 (512x768x97, S = 4992, full 2B DiT + VAE decoder, synthetic weights seeds 31/32)",
                        "args": repr(args), "latents_sub": "[:, ::8] of the final [1,4992,128] latents", "video_slice": "[:, :, ::8, ::16, ::16] of the [1,3,97,512,768] post-processed video"})


def c3_case():
    """BASELINE config C3's PRESET at C1's geometry: LTX-Video 0.9.5 (configs.rs:163-184: 40 steps on the linspace schedule
    with the resolution-dependent shift, CFG 3.0 + STG 1.0 through skip block 19, rescale 0.7, no decode timestep / noise)
    on the full 2B DiT and VAE decoder (weights of c1_case), 256x384x25: 120 real-width forwards through the guidance
    path (t2v_pipeline.rs:878-964), which the toy pipeline fixture only walks with 2-layer models.  f32.
    Committed: final latents, the latents after steps 1 and 20, a strided slice + moments of the video."""
    import time
    dcfg, vcfg = O.DitConfig(), O.VaeConfig()
    dw = O.synth_weights(O.dit_weight_shapes(dcfg), seed=31)
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=32)
    lat, pe, pm, noise, mean, std = c1_inputs()
    ne = torch.randn(1, 128, 4096, generator=torch.Generator().manual_seed(43))
    nm = torch.zeros(1, 128); nm[:, :8] = 1
    args = O.PipelineArgs(height=C1["height"], width=C1["width"], num_frames=C1["num_frames"], num_inference_steps=40, sigmas=None,
                          guidance_scale=3.0, guidance_rescale=0.7, stg_scale=1.0, skip_block_list=[19], decode_timestep=0.0, decode_noise_scale=0.0)
    traj = []
    t0 = time.time()
    video = O.pipeline_call(dw, dcfg, vw, vcfg, mean, std, args, lat, pe, pm, ne, nm, noise, torch.float32, trajectory=traj)
    dt = time.time() - t0
    out = {"dit_weights_checksum": weights_checksum(dw), "vae_weights_checksum": weights_checksum(vw), "latents": traj[-1], "latents_step1": traj[0],
           "latents_step20": traj[19], "video_slice": video[:, :, ::4, ::8, ::8],
           "video_moments": torch.tensor([float(video.double().mean()), float(video.double().std()), float(video.double().abs().sum())], dtype=torch.float64),
           "oracle_seconds": torch.tensor([dt], dtype=torch.float64)}
    print(f"C3-preset oracle: {dt:.1f} s, {len(traj)} steps, video mean {float(video.mean()):.2f} std {float(video.std()):.2f}", flush=True)
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, "oracle_c3.safetensors"),
              metadata={"source": "oracle/ltx_oracle.py pipeline_call, 0.9.5 preset (CFG 3.0 + STG 1.0 skip block 19, rescale 0.7, 40 steps) at C1 geometry, synthetic weights seeds 31/32",
                        "args": repr(args), "video_slice": "[:, :, ::4, ::8, ::8] of the [1,3,25,256,384] post-processed video"})


def c4_case():
    """BASELINE config C4's decode at FULL size: the full VAE decoder (weights of c1_case) on a C2-geometry latent
    [1,128,13,16,24] -> [1,3,97,512,768], once untiled (vae.rs:2101-2136 direct path, 48.4 TFLOP) and once with the
    reference's tiled framewise decode at its default tile parameters (vae.rs:2225-2290, 2358-2434: 2 x 2 spatial x 13
    temporal tiles, blends), f32, decode timestep 0.05.  Committed: a strided slice + moments of each video."""
    import time
    vcfg = O.VaeConfig()
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=32)
    z = torch.randn(1, 128, 13, 16, 24, generator=torch.Generator().manual_seed(46))
    temb = torch.full((1,), 0.05, dtype=torch.float32)
    out = {"vae_weights_checksum": weights_checksum(vw),       # the latents are re-derived from the seed by the test
           "latents_checksum": torch.tensor([float(z.double().sum()), float(z.double().abs().sum())], dtype=torch.float64)}
    for tag, tiling, framewise in (("untiled", False, False), ("tiled", True, True)):
        t0 = time.time()
        v = O.vae_decode(vw, vcfg, z, temb, torch.float32, tiling, framewise)
        dt = time.time() - t0
        out[f"video_slice_{tag}"] = v[:, :, ::8, ::16, ::16]
        out[f"video_edge_{tag}"] = v[:, :, :, 376:392:2, 376:392:2]            # across the spatial tile seam (stride 384) on every frame
        out[f"video_moments_{tag}"] = torch.tensor([float(v.double().mean()), float(v.double().std()), float(v.double().abs().sum())], dtype=torch.float64)
        out[f"oracle_seconds_{tag}"] = torch.tensor([dt], dtype=torch.float64)
        print(f"C4 oracle {tag}: {dt:.1f} s, video {tuple(v.shape)} mean {float(v.mean()):.4f} std {float(v.std()):.4f}", flush=True)
        del v
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, "oracle_c4.safetensors"),
              metadata={"source": "oracle/ltx_oracle.py vae_decode at C2 latent geometry, untiled and tiled+framewise (reference tile parameters), synthetic weights seed 32",
                        "video_slice": "[:, :, ::8, ::16, ::16]; video_edge: [:, :, :, 376:392:2, 376:392:2] of the [1,3,97,512,768] video (before postprocess)"})


def c5vae_case():
    """BASELINE config C5's DECODE geometry: the full VAE decoder (weights of c1_case) on a latent with C5's spatial plane,
    22 x 38 (704 x 1216 pixels) - not a multiple of the 16 x 16 voxel patches the conv kernels tile the plane with at any
    stage (22, 44, 88, 176 rows x 38, 76, 152, 304 columns), so every stage has ragged right and bottom tiles - and 2 latent
    frames (9 video frames; the full 21 frames are 170 TFLOP, hours on the host).  f32, decode timestep 0.05.
    Committed: a strided slice, the right and bottom edge strips on a stride, and moments."""
    import time
    vcfg = O.VaeConfig()
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=32)
    z = torch.randn(1, 128, 2, 22, 38, generator=torch.Generator().manual_seed(47))
    temb = torch.full((1,), 0.05, dtype=torch.float32)
    t0 = time.time()
    v = O.vae_decode(vw, vcfg, z, temb, torch.float32, False, False)
    dt = time.time() - t0
    assert tuple(v.shape) == (1, 3, 9, 704, 1216)
    out = {"vae_weights_checksum": weights_checksum(vw),
           "latents_checksum": torch.tensor([float(z.double().sum()), float(z.double().abs().sum())], dtype=torch.float64),
           "video_slice": v[:, :, ::2, ::16, ::16], "video_right": v[:, :, ::2, ::8, 1184:1216:2], "video_bottom": v[:, :, ::2, 672:704:2, ::8],
           "video_moments": torch.tensor([float(v.double().mean()), float(v.double().std()), float(v.double().abs().sum())], dtype=torch.float64),
           "oracle_seconds": torch.tensor([dt], dtype=torch.float64)}
    print(f"C5-geometry VAE oracle: {dt:.1f} s, video {tuple(v.shape)} mean {float(v.mean()):.4f} std {float(v.std()):.4f}", flush=True)
    save_file({k: c(x) for k, x in out.items()}, os.path.join(GOLD, "oracle_c5vae.safetensors"),
              metadata={"source": "oracle/ltx_oracle.py vae_decode of a [1,128,2,22,38] latent (C5's 704x1216 plane, 2 latent frames), untiled, synthetic weights seed 32",
                        "video_slice": "[:, :, ::2, ::16, ::16]; video_right: [:, :, ::2, ::8, 1184:1216:2]; video_bottom: [:, :, ::2, 672:704:2, ::8] of the [1,3,9,704,1216] video (before postprocess)"})


C3_FULL_SIGMAS = [1.0, 0.8965, 0.6405, 0.1005]     # integer timesteps 1000, 896, 640, 100: exact in bf16 (ltx_transformer.rs:1051)


def c3full_case():
    """BASELINE config C3 at ITS OWN geometry, 512x768x97 (S = 4992): the 0.9.5 preset's guidance path (configs.rs:163-184: CFG 3.0
    + STG 1.0 through skip block 19 + rescale 0.7 = three forwards per step, t2v_pipeline.rs:878-964) for four steps of a
    custom schedule whose timesteps are exact in bf16, + the untiled decode; full 2B DiT + VAE decoder (weights of c1_case),
    f32.  12 forwards at S = 4992 + 48 TFLOP of decode: about 15 minutes of host time.
    Committed: every 8th token of the final latents and of the latents after step 1, moments, a strided slice + moments of the video."""
    import time
    dcfg, vcfg = O.DitConfig(), O.VaeConfig()
    dw = O.synth_weights(O.dit_weight_shapes(dcfg), seed=31)
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=32)
    F, H, W = 13, 16, 24
    lat = O.pack_latents(O.Pcg32(42, 1442695040888963407).randn((1, 128, F, H, W)))
    _, pe, pm, _, mean, std = c1_inputs()
    ne = torch.randn(1, 128, 4096, generator=torch.Generator().manual_seed(43))
    nm = torch.zeros(1, 128); nm[:, :8] = 1
    args = O.PipelineArgs(height=512, width=768, num_frames=97, num_inference_steps=4, sigmas=C3_FULL_SIGMAS,
                          guidance_scale=3.0, guidance_rescale=0.7, stg_scale=1.0, skip_block_list=[19], decode_timestep=0.0, decode_noise_scale=0.0)
    traj = []
    t0 = time.time()
    video = O.pipeline_call(dw, dcfg, vw, vcfg, mean, std, args, lat, pe, pm, ne, nm, None, torch.float32, trajectory=traj)
    dt = time.time() - t0
    l = traj[-1]
    out = {"dit_weights_checksum": weights_checksum(dw), "vae_weights_checksum": weights_checksum(vw),
           "latents_sub": l[:, ::8], "latents_step1_sub": traj[0][:, ::8],
           "latents_moments": torch.tensor([float(l.double().sum()), float(l.double().abs().sum()), float(l.double().pow(2).sum())], dtype=torch.float64),
           "video_slice": video[:, :, ::8, ::16, ::16],
           "video_moments": torch.tensor([float(video.double().mean()), float(video.double().std()), float(video.double().abs().sum())], dtype=torch.float64),
           "oracle_seconds": torch.tensor([dt], dtype=torch.float64)}
    print(f"C3 at 512x768x97 oracle: {dt:.1f} s, {len(traj)} guided steps, video mean {float(video.mean()):.2f} std {float(video.std()):.2f}", flush=True)
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, "oracle_c3_full.safetensors"),
              metadata={"source": "oracle/ltx_oracle.py pipeline_call, 0.9.5 preset's guidance (CFG 3.0 + STG 1.0 skip block 19, rescale 0.7), 4 steps, at 512x768x97 (S = 4992), synthetic weights seeds 31/32",
                        "args": repr(args), "latents_sub": "[:, ::8] of the [1,4992,128] latents", "video_slice": "[:, :, ::8, ::16, ::16] of the [1,3,97,512,768] post-processed video"})


C5_DIT_CFG = dict(in_channels=128, out_channels=128, num_attention_heads=32, attention_head_dim=128, cross_attention_dim=4096,
                  num_layers=2, caption_channels=4096)


def c5dit_inputs():
    """Inputs of the two-layer 13B-width forward on C5's full grid; shared by the generator and tests/test_gpu_c5.py."""
    F, H, W, K = 21, 22, 38, 128
    g = torch.Generator().manual_seed(516)
    hidden = torch.randn(1, F * H * W, 128, generator=g)
    enc = torch.randn(1, K, 4096, generator=g)
    mask = torch.zeros(1, K); mask[:, :45] = 1
    t = torch.tensor([896.0])                        # exact in bf16
    return F, H, W, K, hidden, enc, mask, t


def c5dit_case():
    """BASELINE config C5's DiT at its own launch sizes: a TWO-layer model with the 13B block shape (D = 4096, 32 heads x 128,
    caption / cross-attention dim 4096; configs.rs:243-282) on the FULL 21 x 22 x 38 grid, S = 17556 - the head_dim-128 attention and
    the K = 4096 / 16384 GEMM plans at the sizes the 48-layer model launches them with.  The oracle's attention runs the heads in
    passes (1.2 GB of f32 scores per head).  Two runs: plain f32, and f32 arithmetic on bf16-rounded weights / inputs (the
    reference for the bf16 production kernels).  Committed: every 16th token of each output + moments."""
    import time
    cfg = O.DitConfig(**C5_DIT_CFG)
    w = O.synth_weights(O.dit_weight_shapes(cfg), seed=515)
    F, H, W, K, hidden, enc, mask, t = c5dit_inputs()
    coords = O.build_video_coords(1, F, H, W)
    out = {"dit_weights_checksum": weights_checksum(w)}
    for tag, rnd in (("f32", False), ("bf16in", True)):
        ww = {k: v.bfloat16().float() for k, v in w.items()} if rnd else w
        hh, ee = (hidden.bfloat16().float(), enc.bfloat16().float()) if rnd else (hidden, enc)
        t0 = time.time()
        y = O.dit_forward(ww, cfg, hh, ee, t, mask, F, H, W, None, coords)
        dt = time.time() - t0
        out[f"out_sub_{tag}"] = y[:, ::16]
        out[f"out_moments_{tag}"] = torch.tensor([float(y.double().sum()), float(y.double().abs().sum()), float(y.double().pow(2).sum())], dtype=torch.float64)
        out[f"oracle_seconds_{tag}"] = torch.tensor([dt], dtype=torch.float64)
        print(f"C5 two-layer DiT oracle {tag}: {dt:.1f} s, out std {float(y.std()):.4f}", flush=True)
        del y, ww
    save_file({k: c(v) for k, v in out.items()}, os.path.join(GOLD, "oracle_c5dit.safetensors"),
              metadata={"source": "oracle/ltx_oracle.py dit_forward, 2 layers at 13B width (D 4096, 32 x 128 heads) on the 21x22x38 grid (S = 17556), synthetic weights seed 515",
                        "out_sub": "[:, ::16] of the [1,17556,128] output; bf16in = the same arithmetic on bf16-rounded weights / hidden / enc"})


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "c1":
        c1_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c2":
        c2_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c3":
        c3_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c4":
        c4_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c5vae":
        c5vae_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c3full":
        c3full_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c5dit":
        c5dit_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ref":
        ref_scripts(); ref_scripts_imported(); ref_rope_tables()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rope":
        ref_rope_tables()
        sys.exit(0)
    ref_scripts()
    ref_scripts_imported()
    ref_rope_tables()
    for n, s in DIT_CASES.items():
        dit_case(n, s)
    vae_case()
    ops_case()
    pipeline_case()
    c1_case()
    c3_case()
    c4_case()
    c5vae_case()
    c5dit_case()
    c2_case()           # the headline config in full: ~20 minutes of host time (two 10-minute oracle runs)
    c3full_case()       # ~15 minutes
    tot = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print("fixtures written:", sorted(os.listdir(GOLD)), f"{tot / 1e6:.1f} MB")
