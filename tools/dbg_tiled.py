import sys, torch
sys.path.insert(0,'candle-video_amd'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import ltxhip, ltx_oracle as O
from tools_cfg import VAE_CFG
cfg = O.VaeConfig(**VAE_CFG)
w = O.synth_weights(O.vae_decoder_weight_shapes(cfg), seed=7)
m = ltxhip.AutoencoderKLLtxVideo(ltxhip.AutoencoderKLLtxVideoConfig(**VAE_CFG), {"decoder."+k: v.cuda() for k,v in w.items()}, torch.float32)
t = torch.tensor([0.05])
g = torch.Generator().manual_seed(1)
z2 = torch.randn(1,8,2,3,3,generator=g)
for (mn, st) in [(32,32),(64,64),(64,32),(96,64)]:
    tc = O.VaeConfig(**VAE_CFG, tile_sample_min_height=mn, tile_sample_min_width=mn, tile_sample_stride_height=st, tile_sample_stride_width=st)
    ref = O.tiled_decode(w, tc, z2, t, torch.float32)
    m.use_tiling=True; m.tile_sample_min_height=m.tile_sample_min_width=mn; m.tile_sample_stride_height=m.tile_sample_stride_width=st
    out = m.decode(z2.cuda(), t).cpu()
    err = (out-ref).abs().amax(dim=(0,1,2))
    print("min",mn,"stride",st)
    for y in range(0,96,16): print(' '.join(f"{float(err[y:y+16,x:x+16].max()):8.1e}" for x in range(0,96,16)))
del m
