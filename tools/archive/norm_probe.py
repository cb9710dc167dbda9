#!/usr/bin/env python3
"""What bounds the DiT's row norm pass ([4992, 2048] bf16, 41 MB): the pass with and without its modulation operands, against a
plain device copy of the same bytes and a pure elementwise map (torch), back to back in one process."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, ltxhip
from microbench import timeit
S, D = 4992, 2048
x = torch.randn(S, D, device="cuda").bfloat16(); y = torch.empty_like(x)
sc = torch.randn(1, D, device="cuda"); sh = torch.randn(1, D, device="cuda")
res = {}
def t(name, fn):
    res[name] = round(min(timeit(fn, iters=200, warm=10) for _ in range(3)) * 1e3, 2)
os.environ["LTX_ROWNORM_ROWS"] = "0"
t("rownorm_mod_us", lambda: ltxhip.ops.rownorm(x, 0, 1e-6, None, sc, sh, S, 0))
t("rownorm_plain_us", lambda: ltxhip.ops.rownorm(x, 0, 1e-6, None, None, None, S, 0))
y_old = ltxhip.ops.rownorm(x, 0, 1e-6, None, sc, sh, S, 0); y_old_ln = ltxhip.ops.rownorm(x, 1, 1e-6, None, sc, sh, S, 0)
del os.environ["LTX_ROWNORM_ROWS"]
for occ in ("1", "2", "3", "4", "6"):
    os.environ["LTX_ROWNORM_OCC"] = occ
    t(f"rownorm_mod_occ{occ}_us", lambda: ltxhip.ops.rownorm(x, 0, 1e-6, None, sc, sh, S, 0))
del os.environ["LTX_ROWNORM_OCC"]
os.environ["LTX_ROWNORM_ROWS"] = "1"
t("rownorm_rows_mod_us", lambda: ltxhip.ops.rownorm(x, 0, 1e-6, None, sc, sh, S, 0))
t("rownorm_rows_plain_us", lambda: ltxhip.ops.rownorm(x, 0, 1e-6, None, None, None, S, 0))
res["rows_bit_identical"] = bool(torch.equal(y_old, ltxhip.ops.rownorm(x, 0, 1e-6, None, sc, sh, S, 0)) and torch.equal(y_old_ln, ltxhip.ops.rownorm(x, 1, 1e-6, None, sc, sh, S, 0)))
x3 = torch.randn(3 * S - 5, D, device="cuda").bfloat16(); sc3 = torch.randn(3, D, device="cuda"); sh3 = torch.randn(3, D, device="cuda")
os.environ["LTX_ROWNORM_ROWS"] = "0"; y3 = ltxhip.ops.rownorm(x3, 0, 1e-6, None, sc3, sh3, S, 0); os.environ["LTX_ROWNORM_ROWS"] = "1"
res["rows_bit_identical_3_batches_ragged"] = bool(torch.equal(y3, ltxhip.ops.rownorm(x3, 0, 1e-6, None, sc3, sh3, S, 0)))
os.environ.pop("LTX_ROWNORM_ROWS", None)
rs = ltxhip.ops.rowsq(x)
t("rownorm_presum_mod_us", lambda: ltxhip.ops.rownorm_presum(x, rs, 1e-6, None, sc, sh, S, 0))
t("copy_us", lambda: y.copy_(x))
t("map_mul_us", lambda: torch.mul(x, 1.5, out=y))
w = torch.randn(D, device="cuda").bfloat16()
t("qknorm_noRope_us", lambda: ltxhip.ops.qknorm_rope(x, w, 1e-5))
print(json.dumps(res))
