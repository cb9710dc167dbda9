"""CPU suite, part 2: the C-ABI library loads without a GPU, exports every symbol the headers declare,
and its host-side scalar restatements (scheduler, shift, PCG32, video_coords) agree with the oracle.
No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import ltx_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    return ltxhip


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ltx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hip):
    lib = ctypes.CDLL(os.path.join(ROOT, "candle-video_amd", "libltxhip.so"))
    names = _declared("ltxhip.h") + _declared("ltxhip_ops.h") + _declared("ltxhip_weights.h") + _declared("ltxhip_t5.h") + _declared("ltxhip_frames.h") + _declared("ltxhip_presets.h") + _declared("ltxhip_team.h")
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), f"libltxhip.so does not export {n}"
    assert set(hip.EXPORTED_SYMBOLS) == set(names), set(hip.EXPORTED_SYMBOLS) ^ set(names)


def test_default_configs_match_reference_defaults(hip):
    c = hip.DitConfigC(); hip.lib.ltx_dit_config_default(ctypes.byref(c))        # ltx_transformer.rs:40-58
    assert (c.in_channels, c.num_attention_heads, c.attention_head_dim, c.num_layers, c.caption_channels, c.cross_attention_dim) == (128, 32, 64, 28, 4096, 2048)
    v = hip.VaeConfigC(); hip.lib.ltx_vae_config_default(ctypes.byref(v))        # vae.rs:68-103
    assert list(v.decoder_block_out_channels)[:3] == [256, 512, 1024] and v.patch_size == 4 and v.decoder_causal == 0
    t = hip.TilingC(); hip.lib.ltx_tiling_default(ctypes.byref(t))               # vae.rs:1849-1861
    assert (t.tile_sample_min_height, t.tile_sample_stride_height, t.tile_sample_min_num_frames, t.tile_sample_stride_num_frames) == (512, 384, 16, 8)
    p = hip.PipelineParamsC(); hip.lib.ltx_pipeline_params_default(ctypes.byref(p))
    assert p.frame_rate == 25 and p.num_inference_steps == 7 and abs(p.decode_noise_scale - 0.025) < 1e-7


def test_scheduler_and_shift_match_oracle_and_fixture(hip, golden):
    g = golden("oracle_ops.safetensors")
    s = hip.FlowMatchEulerDiscreteScheduler()
    ts = s.set_timesteps([1.0, 0.9937, 0.9875, 0.9812, 0.9750, 0.9094, 0.7250], 0.0)
    assert ts == g["sched_distilled_timesteps"].tolist() == [1000, 979, 959, 938, 918, 703, 99]
    assert np.abs(np.array(s.sigmas, np.float32) - g["sched_distilled_sigmas"].numpy()).max() < 1e-7   # bar MSE < 1e-6 (verify_scheduler_parity.rs:217-228)
    for S in (384, 4992, 17556):
        mu = hip.calculate_shift(S)
        assert mu == O.calculate_shift(S)
        lin = list(O.FlowMatchEulerScheduler._linspace(1.0, 1.0 / 40, 40))
        ts = s.set_timesteps(lin, mu)
        assert ts == g[f"sched40_S{S}_timesteps"].tolist()
        assert np.abs(np.array(s.sigmas, np.float32) - g[f"sched40_S{S}_sigmas"].numpy()).max() < 1e-6


def test_mu_is_monotonic_property(hip):
    # tests/verify_pipeline_parity.rs prop_mu_calculation_monotonic
    mus = [hip.calculate_shift(s) for s in range(64, 20000, 997)]
    assert all(b > a for a, b in zip(mus, mus[1:]))


def test_pcg32_and_coords_match_oracle(hip, golden):
    g = golden("oracle_ops.safetensors")
    a = hip.pcg32_randn(42, (32,))
    assert (a - g["pcg_randn"]).abs().max() < 1e-6            # libm vs numpy log/cos: 1-ulp noise only
    b = hip.pcg32_randn(42, (1, 128, 4, 8, 12))               # C1 latent shape (main.rs:568-604)
    assert b.shape == (1, 128, 4, 8, 12) and abs(float(b.mean())) < 0.02 and abs(float(b.std()) - 1) < 0.02
    assert torch.equal(hip.pcg32_randn(7, (5,)), hip.pcg32_randn(7, (6,))[:5]) and not torch.equal(hip.pcg32_randn(7, (5,)), hip.pcg32_randn(8, (5,)))
    for (F, H, W) in ((1, 1, 1), (4, 8, 12), (13, 16, 24)):
        assert (hip.build_video_coords(F, H, W) - O.build_video_coords(1, F, H, W)[0]).abs().max() < 1e-10   # verify_video_coords_parity.rs:146
    assert torch.equal(hip.pack_latents(b), O.pack_latents(b))


def test_pcg32_integer_stream_is_bit_exact(hip, golden):
    """The u32 stream against (1) the PCG paper's demo vector (pcg32 seed 42, sequence 54), (2) the reference's own
    scripts/verify_rng.py Pcg32 executed unmodified (tests/golden/ref_rng.safetensors), (3) the oracle; and the
    Gaussians built on it against the same script (its own bar: 1e-6)."""
    assert hip.pcg32_u32(42, 6, inc=54).tolist() == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    r = golden("ref_rng.safetensors")
    assert hip.pcg32_u32(42, 6, inc=54).tolist() == r["u32_seed42_seq54"].tolist()
    assert torch.equal(hip.pcg32_u32(42, 64), r["u32"])
    o = O.Pcg32(42, 1442695040888963407)
    assert [o.next_u32() for _ in range(64)] == r["u32"].tolist()
    assert (hip.pcg32_randn(42, (257,)) - r["randn"]).abs().max() < 1e-6
    assert (O.Pcg32(42, 1442695040888963407).randn((257,)) - r["randn"]).abs().max() < 1e-6


def test_errors_surface_as_exceptions_without_gpu(hip):
    with pytest.raises(hip.LtxError, match="bad argument"):
        hip.FlowMatchEulerDiscreteScheduler().set_timesteps([], 0.0)
    with pytest.raises(hip.LtxError, match="GPU"):       # host tensors are refused, never silently computed on the CPU
        hip.guidance_combine(torch.zeros(1, 4, 8))


def test_team_entry_points_reject_bad_arguments_without_a_gpu(hip):
    """ltxhip_team.h (RCCL behind the C ABI for hosts without torch.distributed): argument checks come before librccl.so is
    loaded or a device is touched."""
    h = ctypes.c_void_p()
    ident = (ctypes.c_char * 128)()
    assert hip.lib.ltx_team_create(None, 1, 0, 0, ctypes.byref(h)) == 1
    assert hip.lib.ltx_team_create(ident, 2, 2, 0, ctypes.byref(h)) == 1 and b"rank" in hip.lib.ltx_last_error()
    assert hip.lib.ltx_team_create(ident, 0, 0, 0, ctypes.byref(h)) == 1
    assert hip.lib.ltx_team_unique_id(None) == 1
    assert hip.lib.ltx_team_size(None) == 0 and hip.lib.ltx_team_rank(None) == -1
    assert hip.lib.ltx_team_allgather_f32(None, None, None, 4, None) == 1
    assert hip.lib.ltx_team_exchange_f32(None, None, 0, -1, None, 0, -1, None) == 1


def test_team_without_rccl_reports_unsupported_instead_of_crashing():
    """A host without librccl.so: the first ltx_team_* call must return LTX_ERR_UNSUPPORTED (4) with the loader's message
    (the loader's dlerror() text is read once - reading it twice hands std::string a null pointer). The library is
    loaded once per process, so the branch is walked in a child with LTX_RCCL_LIB naming a file that does not exist."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import ctypes, sys\n"
        f"lib = ctypes.CDLL({os.path.join(root, 'candle-video_amd', 'libltxhip.so')!r})\n"
        "lib.ltx_last_error.restype = ctypes.c_char_p\n"
        "ident = (ctypes.c_char * 128)()\n"
        "rc = lib.ltx_team_unique_id(ident)\n"
        "msg = lib.ltx_last_error()\n"
        "h = ctypes.c_void_p()\n"
        "rc2 = lib.ltx_team_create(ident, 1, 0, 0, ctypes.byref(h))\n"
        "print(rc, rc2, msg.decode())\n")
    env = dict(os.environ, LTX_RCCL_LIB="/nonexistent/librccl-missing.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rc, rc2, msg = r.stdout.strip().split(" ", 2)
    assert (rc, rc2) == ("4", "4") and "cannot load librccl.so" in msg and "librccl-missing" in msg, r.stdout


@pytest.mark.parametrize("kind", ["karras", "exponential", "beta"])
@pytest.mark.parametrize("invert", [False, True])
def test_scheduler_sigma_conversions_match_the_oracle(hip, kind, invert):
    """The options of FlowMatchEulerDiscreteSchedulerConfig no preset enables (scheduler.rs:30-33, 222-272, 363-370, 389-399):
    use_karras_sigmas / use_exponential_sigmas / use_beta_sigmas (the inverse CDF of Beta(0.6, 0.6): statrs there, a continued
    fraction + bisection here, scipy.stats.beta.ppf in the oracle) and invert_sigmas, through ltx_sched_set_timesteps_ex, on the
    distilled list and on linspace schedules of three lengths, with and without mu / stretch."""
    import numpy as np
    import ltx_oracle as O
    flags = {f"use_{kind}_sigmas": True}
    for sig_in, mu, st in (([1.0, 0.9937, 0.9875, 0.9812, 0.9750, 0.9094, 0.7250], 1.3017, 0.1),
                           (list(np.linspace(1.0, 1.0 / 30, 30, dtype=np.float32)), 0.5217, 0.1),
                           (list(np.linspace(1.0, 1.0 / 8, 8, dtype=np.float32)), None, None),
                           (list(np.linspace(1.0, 1.0 / 40, 40, dtype=np.float32)), 3.428, 0.1)):
        ref = O.FlowMatchEulerScheduler(O.SchedulerCfg(shift=1.0, shift_terminal=st, invert_sigmas=invert, **flags))
        want_t = ref.set_timesteps(len(sig_in), sigmas=sig_in, mu=mu)
        got = hip.FlowMatchEulerDiscreteScheduler(shift=1.0, shift_terminal=st, invert_sigmas=invert, **flags)
        got_t = got.set_timesteps(sig_in, mu)
        assert len(got.sigmas) == len(sig_in) + 1 and got.sigmas[-1] == (1.0 if invert else 0.0)
        assert np.abs(np.asarray(got.sigmas, dtype=np.float64) - ref.sigmas.astype(np.float64)).max() < 2e-6, (kind, invert, mu)
        assert all(abs(a - b) <= 1 for a, b in zip(got_t, want_t))             # truncation of values within 2e-3 of each other
    with pytest.raises(hip.LtxError):
        hip.FlowMatchEulerDiscreteScheduler(use_karras_sigmas=True, use_beta_sigmas=True)


def test_scheduler_beta_ppf_known_answers(hip):
    """Beta(0.6, 0.6) is symmetric: ppf(0.5) = 0.5, ppf(1 - t) = 1 - ppf(t); ends 0 and 1 (scheduler.rs:260-270 evaluates
    1 - linspace(0, 1, n), i.e. both ends)."""
    s = hip.FlowMatchEulerDiscreteScheduler(shift=1.0, shift_terminal=None, use_beta_sigmas=True)
    s.set_timesteps([1.0, 0.75, 0.5, 0.25, 0.0625], None)
    smax, smin = s.sigmas[0], s.sigmas[4]
    p = [(v - smin) / (smax - smin) for v in s.sigmas[:5]]
    assert p[0] == 1.0 and p[4] == 0.0 and abs(p[2] - 0.5) < 1e-6 and abs(p[1] + p[3] - 1.0) < 1e-6
