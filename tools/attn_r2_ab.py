#!/usr/bin/env python3
"""Same-box, same-process A/B of a self-attention kernel against an older revision of itself.

  build REV [NAME]   (build container) compile REV's csrc/attn_q64.hip + attn_q64_loop.inc against the current headers and
                     link tools/variants/libltxhip_NAME.so from the current build's other objects
  run NAME [...]     (GPU box) load the shipped library and every variant with dlopen in ONE process, launch the DiT's
                     self-attention shape (S 4992 x 4992, 32 heads x 64, q prescaled) through each in interleaved rounds,
                     isolated (sync between launches' timing groups) and sustained (196 back-to-back launches, the count of
                     one video); check the outputs bit for bit.  One JSON line per library + a verdict line."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd")
VAR = os.path.join(ROOT, "tools", "variants")
HIPCC = "/opt/rocm/bin/hipcc"


def build(rev, name):
    os.makedirs(VAR, exist_ok=True)
    bdir = os.path.join(PKG, "build", "var"); os.makedirs(bdir, exist_ok=True)
    src, inc = os.path.join(bdir, f"attn_q64_{name}.hip"), os.path.join(bdir, f"attn_q64_loop_{name}.inc")
    for path, out in (("candle-video_amd/csrc/attn_q64.hip", src), ("candle-video_amd/csrc/attn_q64_loop.inc", inc)):
        with open(out, "w") as f:
            f.write(subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{path}"], check=True, capture_output=True, text=True).stdout)
    with open(src) as f:
        text = f.read()
    if "ltx_attention_q64_fits" not in text:                       # the guard the current dispatcher asks for (added after round 2)
        with open(src, "a") as f:
            f.write("\nbool ltx_attention_q64_fits(const AttnArgs& a) { return true; }\n")
    obj = os.path.join(bdir, f"attn_q64_{name}.o")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I" + os.path.join(PKG, "csrc"),
                    f'-DQ64_LOOP_INC="{inc}"', "-x", "hip", "-c", src, "-o", obj], check=True)
    objs = []
    for sub in ("csrc", "host"):
        d = os.path.join(PKG, "build", sub)
        objs += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".o") and not f.startswith("attn_q64.")]
    out = os.path.join(VAR, f"libltxhip_{name}.so")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [obj, "-lz", "-ldl"], check=True)
    print("built", out)


def run(names):
    import torch
    libs = {"head": os.path.join(PKG, "libltxhip.so")}
    for n in names:
        libs[n] = os.path.join(VAR, f"libltxhip_{n}.so")
    h = {k: C.CDLL(p, mode=os.RTLD_LOCAL | os.RTLD_NOW) for k, p in libs.items()}
    heads, S, D = 32, 4992, 2048
    g = torch.Generator(device="cuda").manual_seed(5)
    q, k, v = [torch.randn(1, S, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
    qp = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
    outs = {n: torch.empty_like(q) for n in h}
    vp = lambda t: C.c_void_p(t.data_ptr())

    def launch(n):
        rc = h[n].ltx_op_attention_prescaled(vp(qp), vp(k), vp(v), vp(outs[n]), 1, S, S, heads, 64, D, D, D, D, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, n

    def timed(n, iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            launch(n)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    for n in h:
        for _ in range(5):
            launch(n)
    torch.cuda.synchronize()
    res = {n: {"isolated_us": [], "sustained_us": []} for n in h}
    order = list(h)
    for rnd in range(6):
        for n in (order if rnd % 2 == 0 else order[::-1]):
            res[n]["isolated_us"].append(timed(n, 20))
    for rnd in range(4):
        for n in (order if rnd % 2 == 0 else order[::-1]):
            res[n]["sustained_us"].append(timed(n, 196 * 3))
    fl = 4 * heads * S * S * 64
    med = lambda x: sorted(x)[len(x) // 2]
    for n in h:
        r = res[n]
        print(json.dumps({"lib": n, "isolated_us": round(med(r["isolated_us"]), 1), "sustained_us": round(med(r["sustained_us"]), 1),
                          "sustained_TF": round(fl / med(r["sustained_us"]) / 1e6, 1), "frac_of_2.5PF": round(fl / med(r["sustained_us"]) / 2.5e9, 3),
                          "all_isolated": [round(x, 1) for x in r["isolated_us"]], "all_sustained": [round(x, 1) for x in r["sustained_us"]],
                          "bit_identical_to_head": bool(torch.equal(outs[n], outs["head"]))}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "attn_" + sys.argv[2])
    else:
        run(sys.argv[2:])
