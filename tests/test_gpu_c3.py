"""GPU suite: BASELINE config C3's PRESET at real model width - LTX-Video 0.9.5 (configs.rs:163-184): 40 steps on the
shifted linspace schedule, CFG 3.0 + STG 1.0 through skip block 19 + rescale 0.7, i.e. THREE forwards per step through the
guidance path of LtxPipeline::call (t2v_pipeline.rs:878-964), on the full 2B DiT (28 layers, D = 2048) and VAE decoder at
C1's geometry (256x384x25; the CPU oracle needs ~10 minutes for these 120 forwards, C2's geometry would need days).
Fixture: tests/golden/oracle_c3.safetensors (tools/gen_fixtures.py c3), weights seeded and re-derived here.

Bars: f32 mode rel-max <= 1e-3 on the final latents and the video slice after 40 guided steps (north_star's parity bar);
bf16 production kernels: video PSNR > 30 dB against the plain f32 oracle (C1 measures 30.4 dB for 7 un-guided steps on the
same bar: it includes the reference's own bf16-timestep quirk, ltx_transformer.rs:1051)."""
import math
import os

import pytest
import torch
from safetensors.torch import load_file

import ltx_oracle as O
from conftest import rel_l2, rel_max
from test_gpu_c1 import checksum, inputs, psnr

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_c3.safetensors")


@pytest.fixture(scope="module")
def c3():
    import ltxhip
    assert torch.cuda.is_available()
    g = load_file(GOLD)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    dw = O.synth_weights(O.dit_weight_shapes(O.DitConfig()), seed=31)
    assert torch.allclose(checksum(dw), g["dit_weights_checksum"], rtol=1e-9), "synthetic DiT weights differ from the generator's"
    vw = O.synth_weights(O.vae_decoder_weight_shapes(O.VaeConfig()), seed=32)
    assert torch.allclose(checksum(vw), g["vae_weights_checksum"], rtol=1e-9)
    return ltxhip, g, dw, vw


def run(hip, dw, vw, dt):
    lat, pe, pm, noise, mean, std = inputs()
    ne = torch.randn(1, 128, 4096, generator=torch.Generator().manual_seed(43))
    nm = torch.zeros(1, 128); nm[:, :8] = 1
    vwd = {"decoder." + k: v.to(DEV) for k, v in vw.items()}
    vwd["latents_mean"] = mean.to(DEV); vwd["latents_std"] = std.to(DEV)
    pre = hip.get_config_by_version("0.9.5")
    dit = hip.LtxVideoTransformer3DModel(pre.transformer, {k: v.to(DEV) for k, v in dw.items()}, dt)
    vae = hip.AutoencoderKLLtxVideo(pre.vae, vwd, dt)
    del vwd
    pipe = hip.LtxPipeline(dit, vae)
    call = pre.pipeline_call(256, 384, 25, postprocess=True)          # what main.rs:585-646 passes for this preset
    assert (call.num_inference_steps, call.guidance_scale, call.stg_scale, list(call.skip_block_list)) == (40, 3.0, 1.0, [19])
    assert call.guidance_rescale == pytest.approx(0.7, rel=1e-6)
    assert call.decode_timestep == 0.0 and call.decode_noise_scale == 0.0 and not call.sigmas
    lat_f, video = pipe.call(call, lat.to(DEV), pe.to(DEV), pm.to(DEV), ne.to(DEV), nm.to(DEV), decode_noise=noise.to(DEV))
    torch.cuda.synchronize()
    out = lat_f.float().cpu(), video.float().cpu()
    del pipe, dit, vae
    torch.cuda.empty_cache()
    return out


def test_c3_preset_f32_mode_vs_oracle(c3):
    hip, g, dw, vw = c3
    lat, video = run(hip, dw, vw, torch.float32)
    assert torch.isfinite(video).all()
    e_lat = rel_max(lat, g["latents"])
    e_vid = rel_max(video[:, :, ::4, ::8, ::8], g["video_slice"])
    print(f"C3 preset f32 vs oracle after 40 guided steps: latents rel-max {e_lat:.2e}, video slice rel-max {e_vid:.2e}")
    assert e_lat <= 1e-3, e_lat
    assert e_vid <= 1e-3, e_vid
    mom = g["video_moments"]
    assert abs(float(video.double().abs().sum()) / float(mom[2]) - 1.0) <= 1e-4
    assert float(mom[1]) > 10.0                      # the synthetic video is not degenerate
    assert rel_l2(g["latents"], g["latents_step20"]) > 1e-2      # and the trajectory was still moving half-way


def test_c3_preset_bf16_production_kernels_vs_f32_oracle(c3):
    hip, g, dw, vw = c3
    lat, video = run(hip, dw, vw, torch.bfloat16)
    assert torch.isfinite(video).all()
    p = psnr(video[:, :, ::4, ::8, ::8], g["video_slice"])
    e = rel_l2(lat, g["latents"])
    print(f"C3 preset bf16 vs plain f32 oracle: latent rel-L2 {e:.4f}, PSNR {p:.1f} dB")
    assert p > 30.0, (p, e)
