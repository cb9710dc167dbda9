import sys, os, ctypes, json, shutil
ROOT="/root/repo"
shutil.copyfile(ROOT+"/tools/variants/libltxhip_xtrace.so", ROOT+"/candle-video_amd/libltxhip.so")
sys.path.insert(0, ROOT+"/candle-video_amd")
import torch, numpy as np, ltxhip
S,H,D,K=4992,32,64,128
g=torch.Generator(device="cuda").manual_seed(0)
q=(torch.randn(1,S,H*D,device="cuda",generator=g)).bfloat16(); k=torch.randn(1,K,H*D,device="cuda",generator=g).bfloat16(); v=torch.randn(1,K,H*D,device="cuda",generator=g).bfloat16()
bias=torch.zeros(1,K,device="cuda")
for _ in range(5): o=ltxhip.ops.attention(q,k,v,H,0.125,key_bias=bias)
torch.cuda.synchronize()
buf=np.zeros(1024*4*8,dtype=np.uint32)
assert ltxhip.lib.ltx_dbg_xtrace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)==0
t=buf.reshape(1024,4,8).astype(np.int64)[:416]
d=np.diff(t,axis=2)/100.0
print("mean us per segment (entry->setup, setup->u0, u0->u1, u1->u2, u2->end...):", np.round(d.mean(axis=(0,1)),2))
print("block span us:", round(float((t[:,:,5].max()-t[:,:,0].min())/100),2), "per-block total:", round(float((t[:,:,5]-t[:,:,0]).mean()/100),2))
