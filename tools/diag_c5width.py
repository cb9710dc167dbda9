#!/usr/bin/env python3
"""Per-op bisect of the 13B-width one-layer test (D 4096, 32 x 128, S 2046, K 128 with 45 valid): each bf16 op against f32 torch."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, ltxhip
import ltx_oracle as O
dev = "cuda"
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
S, D, H, K = 2046, 4096, 32, 128
g = torch.Generator().manual_seed(1)
for name, M, N, Kk, epi in [("qkv", S, 3 * D, D, 0), ("to_out", S, D, D, 2), ("q2", S, D, D, 0), ("o2", S, D, D, 3), ("ff1", S, 4 * D, D, 1), ("ff2", S, D, 4 * D, 2), ("kv2", K, 2 * D, D, 0), ("cap1", K, D, 4096, 1)]:
    x = torch.randn(M, Kk, generator=g).bfloat16().to(dev); w = (torch.randn(N, Kk, generator=g) / math.sqrt(Kk)).bfloat16().to(dev); b = torch.randn(N, generator=g).bfloat16().to(dev)
    r = torch.randn(M, N, generator=g).bfloat16().to(dev); gt = torch.randn(1, N, generator=g).to(dev)
    y = ltxhip.ops.linear(x, w, b, epi=epi, resid=r if epi >= 2 else None, gate=gt if epi == 2 else None, rows_per_batch=M)
    ref = x.float() @ w.float().T + b.float()
    if epi == 1: ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if epi == 2: ref = r.float() + gt * ref
    if epi == 3: ref = r.float() + ref
    print(name, "plan", ltxhip.ops.gemm_plan(M, N, Kk), "rel", round(rel(y, ref), 5), flush=True)
# segmented qkv
x = torch.randn(S, D, generator=g).bfloat16().to(dev); w = (torch.randn(3 * D, D, generator=g) / 64).bfloat16().to(dev); b = torch.randn(3 * D, generator=g).bfloat16().to(dev)
y = ltxhip.ops.linear_segmented(x, w, b, D)
ref = (x.float() @ w.float().T + b.float()).view(S, 3, D).permute(1, 0, 2)
print("qkv segmented rel", round(rel(y, ref), 5), flush=True)
# self attention
q = (torch.randn(1, S, D, generator=g) * (128 ** -0.5) * math.log2(math.e)).bfloat16().to(dev); k = torch.randn(1, S, D, generator=g).bfloat16().to(dev); v = torch.randn(1, S, D, generator=g).bfloat16().to(dev)
o = ltxhip.ops.attention_prescaled(q, k, v, H)
qs = q[0].float().view(S, H, 128); ks = k[0].float().view(S, H, 128); vs = v[0].float().view(S, H, 128)
p = torch.softmax(torch.einsum("qhd,khd->hqk", qs, ks) * math.log(2.0), -1)
ref = torch.einsum("hqk,khd->qhd", p, vs).reshape(S, D)
print("self attention rel", round(rel(o[0], ref), 5), "finite", bool(torch.isfinite(o.float()).all()), flush=True)
# cross attention with bias
q = torch.randn(1, S, D, generator=g).bfloat16().to(dev); k = torch.randn(1, K, D, generator=g).bfloat16().to(dev); v = torch.randn(1, K, D, generator=g).bfloat16().to(dev)
bias = torch.zeros(1, K); bias[:, 45:] = -10000.0; bias = bias.to(dev)
o = ltxhip.ops.attention(q, k, v, H, 128 ** -0.5, bias)
att = torch.einsum("qhd,khd->hqk", q[0].float().view(S, H, 128), k[0].float().view(K, H, 128)) * 128 ** -0.5 + bias[0][None, None, :]
ref = torch.einsum("hqk,khd->qhd", torch.softmax(att, -1), v[0].float().view(K, H, 128)).reshape(S, D)
print("cross attention rel", round(rel(o[0], ref), 5), flush=True)
# q/k norm + rope, D 4096
coords = O.build_video_coords(1, 3, 22, 31)
c, s = ltxhip.ops.rope_table(1, 3, 22, 31, D, coords=coords[0].to(dev))
x = torch.randn(S, D, generator=g).bfloat16().to(dev); w = (1 + 0.1 * torch.randn(D, generator=g)).bfloat16().to(dev)
y = ltxhip.ops.qknorm_rope(x, w, 1e-5, c, s)
ref = O.apply_rotary_emb(O.rms_norm(x.float().cpu()[None], w.float().cpu(), 1e-5), c.cpu().repeat_interleave(2, -1)[None], s.cpu().repeat_interleave(2, -1)[None])[0]
print("qknorm_rope rel", round(rel(y.cpu(), ref), 5), flush=True)
# rownorm + adaLN
x = torch.randn(S, D, generator=g).bfloat16().to(dev); sc = torch.randn(1, D, generator=g).to(dev); sh = torch.randn(1, D, generator=g).to(dev)
y = ltxhip.ops.rownorm(x, 0, 1e-6, None, sc, sh, S, 0)
ref = O.rms_norm(x.float().cpu()[None], None, 1e-6)[0] * (1 + sc.cpu()) + sh.cpu()
print("rownorm rel", round(rel(y.cpu(), ref), 5), flush=True)
# whole model, modes compared layer-free: proj_in only etc. is covered by the ops above; finally the model in both dtypes vs each other
cfgd = dict(in_channels=128, out_channels=128, num_attention_heads=32, attention_head_dim=128, cross_attention_dim=4096, num_layers=1, caption_channels=4096)
w = O.synth_weights(O.dit_weight_shapes(O.DitConfig(**cfgd)), seed=513)
gg = torch.Generator().manual_seed(514)
hidden = torch.randn(1, S, 128, generator=gg); enc = torch.randn(1, K, 4096, generator=gg)
mask = torch.zeros(1, K); mask[:, :45] = 1
for variant, env in (("default", {}), ("no asm16", {"gemm_off": "asm16"}), ("q128 off", {"attn_off": "q128"}), ("no tune", {"gemm_tune": "0"}), ("gemm_big off", {"gemm_off": "big"})):
    for k2, v2 in env.items(): ltxhip.set_option(k2, v2)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        m = ltxhip.LtxVideoTransformer3DModel(ltxhip.LtxVideoTransformer3DModelConfig(**cfgd), {kk: vv.to(dev) for kk, vv in w.items()}, dt)
        outs[dt] = m.forward(hidden.to(dev), enc.to(dev), torch.tensor([896.0]), mask.to(dev), 3, 22, 31, None, coords.to(dev)).float().cpu()
        del m
    for k2 in env: ltxhip.set_option(k2, None)
    print("model", variant, "bf16 vs f32 mode rel", round(rel(outs[torch.bfloat16], outs[torch.float32]), 5), flush=True)
