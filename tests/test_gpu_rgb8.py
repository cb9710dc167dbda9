"""GPU suite: RGB8 frames from the decoder's last epilogue (ltx_vae_decode / ltx_pipeline_params postprocess = 2).

The reference's CLI turns the post-processed f32 video into u8 frames on the host (examples/ltx-video/main.rs:653-675: permute to
[F, H, W, 3], clamp, truncating cast).  With postprocess = 2 conv_out's unpatchify epilogue writes those bytes itself - a quarter of
the bytes stored, no conversion pass.  Bar: byte-equal to ltx_video_to_rgb8 on the postprocess = 1 video, in both modes, untiled and
tiled (the tiled decode blends f32 tiles and converts behind them), through the decoder and through the one-call pipeline."""
import pytest
import torch

import ltx_oracle as O
from tools_cfg import PIPE_DIT_CFG, VAE_CFG

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tiled", [False, True])
def test_decoder_rgb8_equals_the_frame_conversion(hip, dt, tiled):
    vcfg = O.VaeConfig(**VAE_CFG)
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=81)
    vae = hip.AutoencoderKLLtxVideo(hip.AutoencoderKLLtxVideoConfig(**VAE_CFG), {"decoder." + k: v.to(DEV) for k, v in vw.items()}, dt)
    if tiled:
        vae.use_tiling = vae.use_framewise_decoding = True          # shrunken tile parameters: 2 x 2 spatial x 3 temporal tiles on this latent
        vae.tile_sample_min_height = vae.tile_sample_min_width = 128; vae.tile_sample_stride_height = vae.tile_sample_stride_width = 96
        vae.tile_sample_min_num_frames = 16; vae.tile_sample_stride_num_frames = 8
    z = 1.5 * torch.randn(2, 8, 4, 5, 7, generator=torch.Generator().manual_seed(82))
    t = torch.tensor([0.05, 0.05])
    video = vae.decode(z.to(DEV), t, postprocess=True)
    want = hip.video_to_rgb8(video)
    got = vae.decode(z.to(DEV), t, rgb8=True)
    torch.cuda.synchronize()
    assert got.dtype == torch.uint8 and tuple(got.shape) == (2, video.shape[2], video.shape[3], video.shape[4], 3)
    assert torch.equal(got, want)
    assert 20 < float(got.float().std())                       # a live picture, not a clamped constant


def test_pipeline_call_rgb8(hip):
    dcfg, vcfg = O.DitConfig(**PIPE_DIT_CFG), O.VaeConfig(**VAE_CFG)
    dw = O.synth_weights(O.dit_weight_shapes(dcfg), seed=83)
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=84)
    g = torch.Generator().manual_seed(85)
    F, H, W = 2, 2, 3
    lat = O.pack_latents(O.Pcg32(42, 1442695040888963407).randn((1, 8, F, H, W)))
    pe = torch.randn(1, 16, 32, generator=g); pm = torch.zeros(1, 16); pm[:, :9] = 1
    dit = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(**PIPE_DIT_CFG), {k: v.to(DEV) for k, v in dw.items()}, torch.float32)
    vae = hip.AutoencoderKLLtxVideo(hip.AutoencoderKLLtxVideoConfig(**VAE_CFG), {"decoder." + k: v.to(DEV) for k, v in vw.items()}, torch.float32)
    pipe = hip.LtxPipeline(dit, vae)
    kw = dict(height=64, width=96, num_frames=9, num_inference_steps=2, sigmas=[1.0, 0.6])
    _, video = pipe.call(hip.PipelineCall(**kw), lat.to(DEV), pe.to(DEV), pm.to(DEV))
    _, frames = pipe.call(hip.PipelineCall(output_rgb8=True, **kw), lat.to(DEV), pe.to(DEV), pm.to(DEV))
    torch.cuda.synchronize()
    assert frames.dtype == torch.uint8 and tuple(frames.shape) == (1, 9, 64, 96, 3)
    assert torch.equal(frames, hip.video_to_rgb8(video))
