"""CPU suite, part 6: the reference's FILE-FREE property tests, replayed on the oracle and on the host-side C ABI.

The reference's `verify_*_parity.rs` tests need fixtures that are absent, but a handful are pure properties
(SURVEY.md §8c): they are restated here 1:1 (same ranges, same thresholds) with hypothesis, against
  * the oracle (`oracle/ltx_oracle.py`), and
  * the host functions of libltxhip that need no GPU (`ltx_calculate_shift`, `ltx_sched_set_timesteps`,
    `ltx_pcg32_randn`, `ltx_build_video_coords`), which must agree with the oracle on every drawn case."""
import math

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

import ltx_oracle as O
import ltxhip

PROP = settings(max_examples=100, deadline=None)


def mse(a, b):
    return float(((a.double() - b.double()) ** 2).mean())


@PROP
@given(st.integers(1, 2), st.integers(64, 128), st.integers(1, 4), st.integers(4, 16), st.integers(4, 16))
def test_prop_latent_packing_roundtrip(b, c, f, h, w):                       # verify_pipeline_parity.rs:742-766
    x = torch.randn(b, c, f, h, w)
    assert mse(O.unpack_latents(O.pack_latents(x), f, h, w), x) < 1e-10
    assert torch.equal(ltxhip.pack_latents(x), O.pack_latents(x))


@PROP
@given(st.integers(256, 2047), st.integers(2048, 4095))
def test_prop_mu_calculation_monotonic(s1, s2):                              # :768-797
    for fn in (O.calculate_shift, ltxhip.calculate_shift):
        mu1, mu2 = fn(s1), fn(s2)
        assert mu2 >= mu1 and 0.5 <= mu1 <= 1.15 and 0.5 <= mu2 <= 1.15
    assert abs(O.calculate_shift(s1) - ltxhip.calculate_shift(s1)) < 1e-6


@PROP
@given(st.integers(1, 2), st.integers(1, 4), st.integers(4, 8), st.integers(4, 8))
def test_prop_latent_normalization_roundtrip(b, f, h, w):                    # :799-828
    x = torch.randn(b, 128, f, h, w)
    mean, std = torch.randn(128) * 0.1, (1 + torch.randn(128) * 0.1).abs()
    assert mse(O.denormalize_latents(O.normalize_latents(x, mean, std, 1.0), mean, std, 1.0), x) < 1e-5


@PROP
@given(st.floats(1.0, 20.0))
def test_prop_cfg_formula_correctness(gs):                                   # :830-855
    u, t = torch.randn(1, 100, 64), torch.randn(1, 100, 64)
    got = O.guidance_combine(t, u, None, gs, 0.0, 0.0)
    alt = u * (1.0 - gs) + t * gs
    assert mse(got, alt) < 1e-6


@PROP
@given(st.floats(0.5, 2.5), st.floats(0.001, 0.999))
def test_prop_time_shift_bounds(mu, t):                                      # verify_scheduler_parity.rs:765-790
    emu = math.exp(mu); base = 1.0 / t - 1.0
    assert 0.0 < emu / (emu + base) < 1.0 and 0.0 < mu / (mu + base) < 1.0
    sched = O.FlowMatchEulerScheduler(O.SchedulerCfg(shift_terminal=None))      # the oracle applies the same shift in set_timesteps
    sched.set_timesteps(sigmas=[t], mu=mu)
    assert abs(float(sched.sigmas[0]) - emu / (emu + base)) < 1e-5


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 999))
def test_prop_step_formula_correctness(seed):                                # :800-858 (40 steps, mu = 1.5)
    sched = O.FlowMatchEulerScheduler(O.SchedulerCfg(shift_terminal=None))
    ts = sched.set_timesteps(sigmas=list(O.FlowMatchEulerScheduler._linspace(1.0, 1.0 / 40, 40)), mu=1.5)
    x = torch.full((1, 128, 2, 4, 4), seed / 1000.0); v = torch.full((1, 128, 2, 4, 4), (seed + 1.0) / 1000.0)
    dt = np.float32(sched.sigmas[1]) - np.float32(sched.sigmas[0])
    assert mse(sched.step(v, float(ts[0]), x), x + v * float(dt)) < 1e-10


@settings(max_examples=30, deadline=None)
@given(st.integers(2, 60), st.sampled_from([384, 4992, 17556]), st.booleans())
def test_host_scheduler_matches_oracle(n, S, terminal):
    """ltx_sched_set_timesteps (host C++) vs the oracle's scheduler: sigmas within 1 ulp-ish, integer timesteps equal."""
    cfg = O.SchedulerCfg(shift_terminal=0.1 if terminal else None)
    so = O.FlowMatchEulerScheduler(cfg)
    sig = list(O.FlowMatchEulerScheduler._linspace(1.0, 1.0 / n, n))
    mu = O.calculate_shift(S)
    to = so.set_timesteps(sigmas=sig, mu=mu)
    sh = ltxhip.FlowMatchEulerDiscreteScheduler(1.0, 0.1 if terminal else None)
    th = sh.set_timesteps([float(x) for x in sig], mu)
    assert np.allclose(np.asarray(sh.sigmas, dtype=np.float64), np.asarray(so.sigmas, dtype=np.float64), rtol=0, atol=2e-6)
    diff = [abs(int(a) - int(b)) for a, b in zip(th, to)]
    assert max(diff) <= 1          # truncation of x.9999997 vs (x+1).0000002 may differ by one count at most
    assert sum(d != 0 for d in diff) <= max(1, n // 20)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2 ** 40), st.integers(1, 300))
def test_host_pcg32_matches_oracle(seed, n):
    got = ltxhip.pcg32_randn(seed, (n,))
    want = O.Pcg32(seed, 1442695040888963407).randn((n,))
    assert torch.allclose(got, want, atol=1e-6, rtol=0)


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 6), st.integers(1, 6), st.integers(1, 6), st.sampled_from([24, 25, 30]))
def test_host_video_coords_match_oracle(F, H, W, fps):
    got = ltxhip.build_video_coords(F, H, W, fps)
    want = O.build_video_coords(1, F, H, W, fps, 8, 32)[0]
    assert torch.allclose(got, want, atol=1e-6)


def test_skip_layer_mask_and_skip_block_logic_shapes():                      # ltx_transformer.rs:1225-1301 (shape-level)
    cfg = O.DitConfig(in_channels=8, out_channels=8, num_attention_heads=2, attention_head_dim=16, cross_attention_dim=32,
                      num_layers=3, caption_channels=32)
    w = O.synth_weights(O.dit_weight_shapes(cfg), seed=1)
    g = torch.Generator().manual_seed(0)
    x, e = torch.randn(2, 12, 8, generator=g), torch.randn(2, 5, 32, generator=g)
    t, m = torch.tensor([500.0, 500.0]), torch.ones(2, 5)
    base = O.dit_forward(w, cfg, x, e, t, m, 2, 2, 3, None, None, None, (), torch.float32)
    slm = torch.zeros(3, 2); slm[1, 0] = 1.0                                  # skip layer 1 for batch row 0 only
    got = O.dit_forward(w, cfg, x, e, t, m, 2, 2, 3, None, None, slm, (), torch.float32)
    assert got.shape == base.shape == (2, 12, 8)
    assert torch.allclose(got[1], base[1], atol=1e-6) and not torch.allclose(got[0], base[0], atol=1e-4)
    allm = torch.zeros(3, 2); allm[1] = 1.0                                   # mask of ones == skip_block_list (exact identity)
    assert torch.allclose(O.dit_forward(w, cfg, x, e, t, m, 2, 2, 3, None, None, allm, (), torch.float32),
                          O.dit_forward(w, cfg, x, e, t, m, 2, 2, 3, None, None, None, (1,), torch.float32), atol=1e-6)
