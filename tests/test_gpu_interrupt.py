"""GPU suite: LtxPipeline::interrupt and the per-step hook in the one-call form (ltx_pipeline_params::interrupt / on_step).

The reference's loop reads `self.interrupt` before every step and skips the step while it is set (`continue`,
t2v_pipeline.rs:861-863); the latents reached so far are still decoded.  `current_timestep` (:865) is what the hook sees."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def rig():
    import ltxhip
    import ltx_oracle as O
    from tools_cfg import PIPE_DIT_CFG, VAE_CFG
    dcfg, vcfg = O.DitConfig(**PIPE_DIT_CFG), O.VaeConfig(**VAE_CFG)
    dw = O.synth_weights(O.dit_weight_shapes(dcfg), seed=21)
    vw = O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=22)
    g = torch.Generator().manual_seed(23)
    F, H, W = 2, 2, 3
    lat = O.pack_latents(O.Pcg32(42, 1442695040888963407).randn((1, 8, F, H, W)))
    pe = torch.randn(1, 16, 32, generator=g); pm = torch.zeros(1, 16); pm[:, :9] = 1
    sig = [1.0, 0.9, 0.7, 0.45, 0.2]
    dit = ltxhip.LtxVideoTransformer3DModel(ltxhip.LtxVideoTransformer3DModelConfig(**PIPE_DIT_CFG), {k: v.to(DEV) for k, v in dw.items()}, torch.float32)
    vae = ltxhip.AutoencoderKLLtxVideo(ltxhip.AutoencoderKLLtxVideoConfig(**VAE_CFG), {"decoder." + k: v.to(DEV) for k, v in vw.items()}, torch.float32)
    pipe = ltxhip.LtxPipeline(dit, vae)
    call = ltxhip.PipelineCall(height=64, width=96, num_frames=9, num_inference_steps=len(sig), sigmas=sig)
    oargs = O.PipelineArgs(height=64, width=96, num_frames=9, num_inference_steps=len(sig), sigmas=sig)
    mean, std = torch.zeros(8), torch.ones(8)

    def oracle(interrupt_at=None):
        return O.pipeline_call(dw, dcfg, vw, vcfg, mean, std, oargs, lat, pe, pm, None, None, None, torch.float32, interrupt_at=interrupt_at)
    return ltxhip, O, pipe, call, (lat.to(DEV), pe.to(DEV), pm.to(DEV)), oracle, sig


def test_hook_sees_every_step_and_changes_nothing(rig):
    hip, O, pipe, call, dev_in, oracle, sig = rig
    lat0, vid0 = pipe.call(call, *dev_in)
    assert pipe.last_steps == (len(sig), len(sig))
    seen = []
    lat1, vid1 = pipe.call(call, *dev_in, on_step=lambda i, n, t: seen.append((i, n, t)) and False)
    torch.cuda.synchronize()
    assert torch.equal(lat0, lat1) and torch.equal(vid0, vid1)
    ts = O.FlowMatchEulerScheduler(O.SchedulerCfg()).set_timesteps(sigmas=list(sig), mu=0.0)
    assert seen == [(i, len(sig), int(t)) for i, t in enumerate(ts)]          # truncated timesteps (scheduler.rs:659)
    assert pipe.last_steps == (len(sig), len(sig))


@pytest.mark.parametrize("k", [0, 2, 4])
def test_hook_stop_skips_the_remaining_steps_and_still_decodes(rig, k):
    hip, O, pipe, call, dev_in, oracle, sig = rig
    calls = []
    lat, vid = pipe.call(call, *dev_in, on_step=lambda i, n, t: calls.append(i) or i >= k)
    torch.cuda.synchronize()
    assert calls == list(range(k + 1))                     # not called again once it has stopped the loop
    assert pipe.last_steps == (k, len(sig))
    want = oracle(interrupt_at=k)
    assert float((vid.cpu() - want).abs().max()) < 0.5     # 0..255 scale, f32 mode
    if k == 0:
        assert torch.equal(lat, dev_in[0])                 # no step ran: the input latents, decoded


def test_interrupt_flag_is_read_before_every_step(rig):
    hip, O, pipe, call, dev_in, oracle, sig = rig
    flag = ctypes.c_int(0)
    # raised by the hook of step 3 (same thread: the hook runs first, then the flag is read), lowered again at step 4: the
    # reference `continue`s only WHILE the flag is set
    def hook(i, n, t):
        flag.value = 1 if i == 3 else 0
        return False
    lat, vid = pipe.call(call, *dev_in, interrupt=flag, on_step=hook)
    torch.cuda.synchronize()
    assert pipe.last_steps == (len(sig) - 1, len(sig))
    flag.value = 1
    lat2, vid2 = pipe.call(call, *dev_in, interrupt=flag)
    assert pipe.last_steps == (0, len(sig)) and torch.equal(lat2, dev_in[0])
    flag.value = 0
    lat3, vid3 = pipe.call(call, *dev_in, interrupt=flag)
    lat0, vid0 = pipe.call(call, *dev_in)
    assert torch.equal(lat3, lat0) and torch.equal(vid3, vid0)


def test_an_exception_in_the_hook_stops_the_loop_and_is_re_raised(rig):
    hip, O, pipe, call, dev_in, oracle, sig = rig
    def hook(i, n, t):
        if i == 1:
            raise ValueError("stop here")
        return False
    with pytest.raises(ValueError, match="stop here"):
        pipe.call(call, *dev_in, on_step=hook)
    assert pipe.last_steps == (1, len(sig))
