#!/usr/bin/env python3
"""Does it matter where a GEMM's operands come from?  Same launch timed (a) back to back (operands in the Infinity Cache),
(b) after a 1-GiB write that evicts them (HBM), (c)/(d) evicted, then the weights / the activations re-read before the launch."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch, ltxhip
S = 4992
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
def timed(fn, pre):
    ts = []
    for _ in range(12):
        pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
for name, N, K, epi in [("qkv", 6144, 2048, 0), ("to_out", 2048, 2048, 0), ("ff1", 8192, 2048, 1), ("ff2", 2048, 8192, 0)]:
    x = torch.randn(S, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    fn = lambda: ltxhip.ops.linear(x, w, b, epi=epi)
    for _ in range(3): fn()
    fl = 2 * S * N * K
    hot = timed(fn, lambda: None)
    cold = timed(fn, lambda: junk.fill_(1))
    def evict_then_touch_w():
        junk.fill_(1); w.sum()
    wwarm = timed(fn, evict_then_touch_w)
    def evict_then_touch_x():
        junk.fill_(1); x.sum()
    xwarm = timed(fn, evict_then_touch_x)
    print(json.dumps({"case": name, "plan": ltxhip.ops.gemm_plan(S, N, K), "TFLOPs": {"all resident": round(fl / hot / 1e9, 1), "all evicted": round(fl / cold / 1e9, 1),
                      "evicted, W re-read first": round(fl / wwarm / 1e9, 1), "evicted, A re-read first": round(fl / xwarm / 1e9, 1)}}))
