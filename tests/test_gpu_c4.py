"""GPU suite: BASELINE config C4's decode at FULL size against the oracle - the full VAE decoder (real widths
128..1024, 45 convs) on a C2-geometry latent [1,128,13,16,24] -> [1,3,97,512,768]:
  * untiled (AutoencoderKLLtxVideo::decode direct path, vae.rs:2101-2136): the 48.4 TFLOP decode the headline metric times;
  * the reference's tiled framewise decode at its default tile parameters (vae.rs:2225-2290, 2358-2434): 2 x 2 spatial x 13
    temporal tiles with the H / W / T blends, which is what bench.py --config c4 times and what the tile-sharded multi-GPU
    form reproduces.
Fixture: tests/golden/oracle_c4.safetensors (tools/gen_fixtures.py c4: a strided slice, a dense strip across the spatial
tile seam on every frame, and moments of each video; weights seeded and re-derived here).
Bars: f32 mode rel-max <= 1e-3; bf16 production kernels (conv_halo / gemm_big / gemm_p8 + fused norms): rel-L2 <= 2e-2
against the f32 oracle and PSNR > 35 dB on the [-1, 1] video mapped to [0, 255]."""
import math
import os

import pytest
import torch
from safetensors.torch import load_file

import ltx_oracle as O
from conftest import rel_l2, rel_max
from test_gpu_c1 import checksum

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_c4.safetensors")


@pytest.fixture(scope="module")
def c4():
    import ltxhip
    assert torch.cuda.is_available()
    g = load_file(GOLD)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    vw = O.synth_weights(O.vae_decoder_weight_shapes(O.VaeConfig()), seed=32)
    assert torch.allclose(checksum(vw), g["vae_weights_checksum"], rtol=1e-9), "synthetic VAE weights differ from the generator's"
    return ltxhip, g, {"decoder." + k: v.to(DEV) for k, v in vw.items()}


def views(v):
    return v[:, :, ::8, ::16, ::16].float().cpu(), v[:, :, :, 376:392:2, 376:392:2].float().cpu()


def psnr_unit(a, b):                      # [-1, 1] video on the [0, 255] scale of the reference's PSNR criterion
    mse = float((((a.double() - b.double()) * 127.5) ** 2).mean())
    return 10.0 * math.log10(255.0 ** 2 / max(mse, 1e-12))


@pytest.mark.parametrize("tag", ["untiled", "tiled"])
def test_c4_full_size_decode_f32_and_bf16_vs_oracle(c4, tag):
    hip, g, vwd = c4
    z = torch.randn(1, 128, 13, 16, 24, generator=torch.Generator().manual_seed(46))
    assert torch.allclose(torch.tensor([float(z.double().sum()), float(z.double().abs().sum())], dtype=torch.float64), g["latents_checksum"], rtol=1e-12)
    z = z.to(DEV)
    for dt in (torch.float32, torch.bfloat16):
        vae = hip.AutoencoderKLLtxVideo(hip.AutoencoderKLLtxVideoConfig(), vwd, dt)
        vae.use_tiling = vae.use_framewise_decoding = (tag == "tiled")
        v = vae.decode(z, [0.05])
        torch.cuda.synchronize()
        assert v.shape == (1, 3, 97, 512, 768) and torch.isfinite(v).all()
        sl, edge = views(v)
        mom = g[f"video_moments_{tag}"]
        if dt == torch.float32:
            e1, e2 = rel_max(sl, g[f"video_slice_{tag}"]), rel_max(edge, g[f"video_edge_{tag}"])
            print(f"C4 {tag} f32 vs oracle: slice rel-max {e1:.2e}, seam strip rel-max {e2:.2e}")
            assert e1 <= 1e-3 and e2 <= 1e-3, (e1, e2)
            assert abs(float(v.double().abs().sum()) / float(mom[2]) - 1.0) <= 1e-4
        else:
            e1, e2 = rel_l2(sl, g[f"video_slice_{tag}"]), rel_l2(edge, g[f"video_edge_{tag}"])
            p = psnr_unit(sl, g[f"video_slice_{tag}"])
            print(f"C4 {tag} bf16 vs f32 oracle: slice rel-L2 {e1:.4f}, seam strip rel-L2 {e2:.4f}, PSNR {p:.1f} dB")
            assert e1 <= 2e-2 and e2 <= 2e-2 and p > 35.0, (e1, e2, p)
        del vae, v
        torch.cuda.empty_cache()


def test_c4_tiled_and_untiled_differ_as_the_reference_says(c4):
    """the tiled path is not a re-tiling of the same function: every tile sees replicate / zero padding at its own borders
    (vae.rs:2246-2257), so the two oracle videos differ away from the origin - the fixture is not two copies of one result"""
    _, g, _ = c4
    assert rel_l2(g["video_slice_tiled"], g["video_slice_untiled"]) > 1e-3
    assert float(g["video_moments_untiled"][1]) > 0.05
