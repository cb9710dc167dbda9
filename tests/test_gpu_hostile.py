"""GPU suite: parity under HOSTILE statistics (VERDICT r5 "weak" 3, "next" 6).

Every other parity test runs on benign synthetic weights (N(0, 1/fan_in), unit norm weights, O(1) activations).  Real
checkpoints are not like that: a few channels carry most of the magnitude, AdaLN tables hold scales of several units, q/k norm
weights spread over two decades, prompts contain outlier tokens.  Here the SAME kernels run on such data against the oracle:
  * linear / conv weights with 1 % of their output channels scaled x50 (heavy-tailed channels),
  * scale_shift_table entries of +-4 on a tenth of the table (large AdaLN scale / shift / gate: ltx_transformer.rs:847-889),
  * norm_q / norm_k weights log-uniform in [0.1, 8] (RmsNorm::forward, ltx_transformer.rs:99-119, feeding the logits of :719-740),
  * a prompt token 1e3 times the others, VAE latents at 5 sigma.
Bars (measured on MI355X, stated where asserted): f32 mode rel-max <= 1e-3 against the oracle everywhere - the north_star bar
holds under these statistics too; bf16 production kernels against the f32 oracle on bf16-rounded weights / inputs: rel-L2
<= 3e-2 (benign data: 0.5e-2 .. 1e-2).
The self-attention's fixed first-tile max (attn_q64.hip) has an exact second pass behind an overflow check: the counter
ltx_attention_fallback_counts says how often it ran - asserted ZERO on benign data and on the hostile-but-plausible set above;
an EXTREME set (every q/k norm weight = 8: logits of sigma ~ 64) makes it fire, and the test reports its rate and cost
(gpurun_out/r6_attn_fallback_report.json -> profiles/)."""
import json
import math
import os
import time

import pytest
import torch

import ltx_oracle as O
from conftest import rel_l2, rel_max
from tools_cfg import PIPE_DIT_CFG, VAE_CFG

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    import ltxhip
    assert torch.cuda.is_available()
    return ltxhip


def hostile(w, seed, norm_w=None):
    """the hostile variant of a synthetic weight set (see the module docstring); norm_w: every q/k norm weight = that value"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, v in w.items():
        v = v.clone()
        if "norm_q" in name or "norm_k" in name:
            v = torch.full_like(v, norm_w) if norm_w is not None else torch.exp(torch.empty_like(v).uniform_(math.log(0.1), math.log(8.0), generator=g))
        elif name.endswith("scale_shift_table"):
            pick = torch.rand(v.shape, generator=g) < 0.1
            sign = torch.where(torch.rand(v.shape, generator=g) < 0.5, -4.0, 4.0)
            v = torch.where(pick, sign, v)
        elif v.dim() >= 2 and name.endswith(".weight"):
            n = v.shape[0]
            k = max(1, n // 100)
            rows = torch.randperm(n, generator=g)[:k]
            v[rows] = v[rows] * 50.0
        out[name] = v
    return out


def dit_pair(hip, cfgd, w, args_cpu, dt):
    m = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(**cfgd), {k: v.to(DEV) for k, v in w.items()}, dt)
    hidden, enc, t, mask, F, H, W, coords = args_cpu
    y = m.forward(hidden.to(DEV), enc.to(DEV), t, mask.to(DEV), F, H, W, None, coords.to(DEV)).float().cpu()
    del m
    torch.cuda.empty_cache()
    return y


def dit_inputs(cfgd, F, H, W, K, seed, B=1, outlier=True):
    g = torch.Generator().manual_seed(seed)
    hidden = torch.randn(B, F * H * W, cfgd["in_channels"], generator=g)
    enc = torch.randn(B, K, cfgd["caption_channels"], generator=g)
    if outlier:
        enc[:, 3] *= 1e3                                        # one token 1e3 times the others (inside the kept keys)
    mask = torch.zeros(B, K); mask[:, : (2 * K) // 3] = 1
    t = torch.tensor([896.0] * B)                              # exact in bf16: both modes see the same timestep (ltx_transformer.rs:1051)
    return hidden, enc, t, mask, F, H, W, O.build_video_coords(B, F, H, W)


@pytest.mark.parametrize("case", ["tiny", "c1_width_384", "c1_width_1560"])
def test_dit_under_hostile_statistics_vs_oracle(hip, case):
    if case == "tiny":
        cfgd = dict(PIPE_DIT_CFG); F, H, W, K, B = 3, 5, 7, 16, 2
    else:
        cfgd = dict(in_channels=128, out_channels=128, num_attention_heads=32, attention_head_dim=64, cross_attention_dim=2048, num_layers=1, caption_channels=4096)
        F, H, W, K, B = (4, 8, 12, 128, 1) if case == "c1_width_384" else (5, 12, 26, 128, 1)
    cfg = O.DitConfig(**cfgd)
    w = hostile(O.synth_weights(O.dit_weight_shapes(cfg), seed=61), seed=62)
    a = dit_inputs(cfgd, F, H, W, K, 63, B)
    hip.attention_fallback_counts(reset=True)
    want = O.dit_forward(w, cfg, *a[:4], F, H, W, None, a[7])
    assert torch.isfinite(want).all() and float(want.std()) > 1e-3
    y32 = dit_pair(hip, cfgd, w, a, torch.float32)
    e32 = rel_max(y32, want)
    wr = {k: v.bfloat16().float() for k, v in w.items()}
    ar = (a[0].bfloat16().float(), a[1].bfloat16().float()) + a[2:]
    want_r = O.dit_forward(wr, cfg, *ar[:4], F, H, W, None, a[7])
    y16 = dit_pair(hip, cfgd, w, a, torch.bfloat16)
    assert torch.isfinite(y32).all() and torch.isfinite(y16).all()
    e16 = rel_l2(y16, want_r)
    fb = hip.attention_fallback_counts()
    print({"case": case, "f32_rel_max": e32, "bf16_rel_l2": round(e16, 5), "out_std": round(float(want.std()), 3), "fallback": fb})
    assert e32 <= 1e-3, e32
    assert e16 <= 3e-2, e16
    assert fb == (0, 0), fb                                     # hostile but plausible: the fixed max holds


def test_vae_decode_of_5_sigma_latents_and_heavy_channels_vs_oracle(hip):
    vcfg = O.VaeConfig(**VAE_CFG)
    w = hostile(O.synth_weights(O.vae_decoder_weight_shapes(vcfg), seed=64), seed=65)
    z = 5.0 * torch.randn(1, 8, 3, 5, 6, generator=torch.Generator().manual_seed(66))
    temb = torch.tensor([0.05])
    want = O.vae_decode(w, vcfg, z, temb, torch.float32, False, False)
    wr = {k: v.bfloat16().float() for k, v in w.items()}
    want_r = O.vae_decode(wr, vcfg, z.bfloat16().float(), temb, torch.float32, False, False)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        vae = hip.AutoencoderKLLtxVideo(hip.AutoencoderKLLtxVideoConfig(**VAE_CFG), {"decoder." + k: v.to(DEV) for k, v in w.items()}, dt)
        res[dt] = vae.decode(z.to(DEV), temb).float().cpu()
        del vae
    e32, e16 = rel_max(res[torch.float32], want), rel_l2(res[torch.bfloat16], want_r)
    print({"vae_hostile_f32_rel_max": e32, "bf16_rel_l2": round(e16, 5), "out_absmax": float(want.abs().max())})
    assert torch.isfinite(res[torch.bfloat16]).all()
    assert e32 <= 1e-3, e32
    assert e16 <= 3e-2, e16


def test_fixed_max_fallback_rate_and_cost(hip):
    """One C1-width layer at S = 1560 (256- and 128-query blocks, 25 key tiles), bf16: benign weights and the hostile set never
    take the exact-max pass; with every q/k norm weight at 8 the logits have sigma ~ 64 and later keys beat the first tile's
    maximum by more than bf16's exponent range in SOME workgroups - the result must still match the oracle, and the counter
    says how many re-ran and what the forward then costs."""
    cfgd = dict(in_channels=128, out_channels=128, num_attention_heads=32, attention_head_dim=64, cross_attention_dim=2048, num_layers=1, caption_channels=4096)
    cfg = O.DitConfig(**cfgd)
    F, H, W, K = 5, 12, 26, 128
    base = O.synth_weights(O.dit_weight_shapes(cfg), seed=61)
    a = dit_inputs(cfgd, F, H, W, K, 63, 1, outlier=False)
    S = F * H * W
    blocks = 32 * ((S // 256) + (1 if S % 256 > 128 else 0) + (1 if 0 < S % 256 <= 128 else 0))
    report = {"shape": {"S": S, "heads": 32, "head_dim": 64, "layers": 1}, "attention_workgroups_upper_bound": blocks, "sets": {}}
    for name, w in (("benign", base), ("hostile", hostile(base, 62)), ("extreme_norm_weights_8", hostile(base, 62, norm_w=8.0))):
        wr = {k: v.bfloat16().float() for k, v in w.items()}
        want_r = O.dit_forward(wr, cfg, a[0].bfloat16().float(), a[1].bfloat16().float(), a[2], a[3], F, H, W, None, a[7])
        m = hip.LtxVideoTransformer3DModel(hip.LtxVideoTransformer3DModelConfig(**cfgd), {k: v.to(DEV) for k, v in w.items()}, torch.bfloat16)
        dev_args = (a[0].to(DEV), a[1].to(DEV), a[2], a[3].to(DEV), F, H, W, None, a[7].to(DEV))
        y = m.forward(*dev_args)                                # warm-up (plans)
        hip.attention_fallback_counts(reset=True)
        y = m.forward(*dev_args)
        torch.cuda.synchronize()
        fb = hip.attention_fallback_counts(reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m.forward(*dev_args)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        e = rel_l2(y.float().cpu(), want_r)
        report["sets"][name] = {"fallback_workgroups_per_forward": fb[0], "forward_ms": round(ms, 4), "bf16_rel_l2_vs_oracle": round(e, 5)}
        assert torch.isfinite(y.float()).all()
        assert e <= 3e-2, (name, e)
        if name != "extreme_norm_weights_8":
            assert fb == (0, 0), (name, fb)
        del m
        torch.cuda.empty_cache()
    print(json.dumps(report))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r6_attn_fallback_report.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert report["sets"]["extreme_norm_weights_8"]["fallback_workgroups_per_forward"] > 0, "the extreme set was meant to overflow the fixed max"
