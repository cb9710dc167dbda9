"""GPU suite: THE headline config as a whole against the oracle - BASELINE C2 = LTX-Video-0.9.8-2B-distilled at 512x768x97
(latent 13x16x24, S = 4992), 7 distilled steps + the untiled 48-TFLOP VAE decode (t2v_pipeline.rs:627-1073,
configs.rs:223-240), full 2B DiT + VAE decoder with seeded synthetic weights, BASELINE.md section 3's synthetic inputs -
exactly what bench.py times.  Fixture: tests/golden/oracle_c2.safetensors (tools/gen_fixtures.py c2: ~10 minutes of host
time per oracle run, which is why only a token-strided subset of the latents, a strided video slice and moments are kept).

Bars (those of C1, tests/test_gpu_c1.py):
  f32 mode  : rel-max <= 1e-3 on the latents subset and the video slice, moments within 1e-4;
  bf16 mode : the production kernels at the sizes the 470 frames/s are measured on - latent rel-L2 <= 2e-2 and video
              PSNR > 35 dB against the oracle fed bf16-rounded timesteps (ltx_transformer.rs:1051), > 30 dB against the
              plain f32 oracle (that difference includes the reference's own timestep quirk)."""
import os

import pytest
import torch
from safetensors.torch import load_file

import ltx_oracle as O
from conftest import rel_l2, rel_max
from test_gpu_c1 import SIGMAS, checksum, inputs, psnr

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_c2.safetensors")
F, H, W = 13, 16, 24


@pytest.fixture(scope="module")
def c2():
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
    _, pe, pm, _, mean, std = inputs()
    lat = O.pack_latents(O.Pcg32(42, 1442695040888963407).randn((1, 128, F, H, W)))
    noise = torch.randn(1, 128, F, H, W, generator=torch.Generator().manual_seed(44))
    vwd = {"decoder." + k: v.to(DEV) for k, v in vw.items()}
    vwd["latents_mean"] = mean.to(DEV); vwd["latents_std"] = std.to(DEV)
    dit = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(), {k: v.to(DEV) for k, v in dw.items()}, dt)
    vae = hip.AutoencoderKLLtxVideo(hip.AutoencoderKLLtxVideoConfig(), vwd, dt)
    del vwd
    pipe = hip.LtxPipeline(dit, vae)
    call = hip.PipelineCall(height=512, width=768, num_frames=97, num_inference_steps=7, sigmas=SIGMAS, guidance_scale=1.0,
                            stg_scale=0.0, decode_timestep=0.05, decode_noise_scale=0.025, postprocess=True)
    lat_f, video = pipe.call(call, lat.to(DEV), pe.to(DEV), pm.to(DEV), decode_noise=noise.to(DEV))
    torch.cuda.synchronize()
    out = lat_f.float().cpu(), video[:, :, ::8, ::16, ::16].float().cpu(), float(video.double().abs().sum())
    del pipe, dit, vae, video
    torch.cuda.empty_cache()
    return out


def test_c2_headline_config_f32_mode_vs_oracle(c2):
    hip, g, dw, vw = c2
    lat, sl, abs_sum = run(hip, dw, vw, torch.float32)
    assert torch.isfinite(lat).all() and torch.isfinite(sl).all()
    e_lat, e_vid = rel_max(lat[:, ::8], g["latents_sub_f32"]), rel_max(sl, g["video_slice_f32"])
    print(f"C2 f32 vs oracle: latents rel-max {e_lat:.2e}, video slice rel-max {e_vid:.2e}")
    assert e_lat <= 1e-3, e_lat
    assert e_vid <= 1e-3, e_vid
    lm = g["latents_moments_f32"]
    assert abs(float(lat.double().abs().sum()) / float(lm[1]) - 1.0) <= 1e-4 and abs(float(lat.double().pow(2).sum()) / float(lm[2]) - 1.0) <= 1e-4
    assert abs(abs_sum / float(g["video_moments_f32"][2]) - 1.0) <= 1e-4
    assert float(g["video_moments_f32"][1]) > 10.0          # the synthetic video is not degenerate


def test_c2_headline_config_bf16_production_kernels_vs_f32_oracle(c2):
    hip, g, dw, vw = c2
    lat, sl, _ = run(hip, dw, vw, torch.bfloat16)
    assert torch.isfinite(lat).all() and torch.isfinite(sl).all()
    e_lat = rel_l2(lat[:, ::8], g["latents_sub_f32_bf16ts"])
    p_q, p_plain = psnr(sl, g["video_slice_f32_bf16ts"]), psnr(sl, g["video_slice_f32"])
    print(f"C2 bf16 vs f32 oracle: latent rel-L2 {e_lat:.4f} (bf16 timesteps), PSNR {p_q:.1f} dB; vs plain f32 oracle PSNR {p_plain:.1f} dB")
    assert e_lat <= 2e-2, e_lat
    assert p_q > 35.0, p_q
    assert p_plain > 30.0, p_plain
