#!/usr/bin/env python3
"""rownorm_presum on [4992, 2048] bf16: plain loads / stores vs streaming (non-temporal) hints on the data, with and without the
modulation operands; kernel time from the launch's own events."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))
import torch, ltxhip
S, D = 4992, 2048
xs = [torch.randn(S, D, device="cuda").bfloat16() for _ in range(4)]
sc = torch.randn(1, D, device="cuda"); sh = torch.randn(1, D, device="cuda")
rs = [ltxhip.ops.rowsq(x) for x in xs]
def run(fn, iters=64):
    for _ in range(8): fn()
    torch.cuda.synchronize(); ltxhip.prof_enable(True)
    for _ in range(iters): fn()
    ms, _, c = ltxhip.prof_report(4); ltxhip.prof_enable(False)
    return round(ms / max(c, 1) * 1e3, 2)
res = {}
i = [0]
for tag, env in (("plain", None), ("nt", "1")):
    if env: os.environ["LTX_NORM_NT"] = env
    else: os.environ.pop("LTX_NORM_NT", None)
    def f_mod():
        k = i[0] % 4; i[0] += 1
        return ltxhip.ops.rownorm_presum(xs[k], rs[k], 1e-6, None, sc, sh, S, 0)
    def f_nomod():
        k = i[0] % 4; i[0] += 1
        return ltxhip.ops.rownorm_presum(xs[k], rs[k], 1e-6, None, None, None, S, 0)
    res[f"{tag}_mod_us"] = run(f_mod); res[f"{tag}_nomod_us"] = run(f_nomod)
    res[f"{tag}_y"] = f_mod()
i[0] = 0; y_nt = res.pop("nt_y"); y_plain = res.pop("plain_y")
os.environ["LTX_NORM_NT"] = "1"; a = ltxhip.ops.rownorm_presum(xs[0], rs[0], 1e-6, None, sc, sh, S, 0)
os.environ.pop("LTX_NORM_NT"); b = ltxhip.ops.rownorm_presum(xs[0], rs[0], 1e-6, None, sc, sh, S, 0)
res["bit_identical"] = bool(torch.equal(a, b))
print(json.dumps(res))
