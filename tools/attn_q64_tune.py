#!/usr/bin/env python3
"""Tuning harness for the generated main loop of csrc/attn_q64.hip.

  build NAME [key=value ...]   generate a loop variant (tools/gen_attn_q64_asm.py options), compile attn_q64.hip against
                               it and link tools/variants/libltxhip_NAME.so from the objects of the current build
  run [NAME ...]               (GPU box) for every variant: install it as the package's library in THIS scratch copy and
                               measure it in a fresh process; prints one JSON line per variant
  measure                      (internal) timings of the library currently installed

Numbers: 'c2' = S 4992 x 4992, 32 heads (the DiT launch) in TFLOP/s; 'big_it' / 'small_it' = microseconds per key tile
of a 256-query / 128-query workgroup (slope of the launch time over the key count at Sq = 4096: two rounds of big or
four rounds of small blocks); 'big_fix' / 'small_fix' = the per-workgroup constant (prologue + epilogue)."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd")
VAR = os.path.join(ROOT, "tools", "variants")
HIPCC = "/opt/rocm/bin/hipcc"


def build(name, opts):
    os.makedirs(VAR, exist_ok=True)
    bdir = os.path.join(PKG, "build", "var"); os.makedirs(bdir, exist_ok=True)
    inc = os.path.join(bdir, f"loop_{name}.inc")
    defs = [o[2:] for o in opts if o.startswith("-D")]
    gopts = [o for o in opts if not o.startswith("-D")]
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_q64_asm.py"), "--out", inc] + gopts, check=True)
    obj = os.path.join(bdir, f"attn_q64_{name}.o")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", f'-DQ64_LOOP_INC="{inc}"'] +
                   [f"-D{d}" for d in defs] + ["-x", "hip", "-c", os.path.join(PKG, "csrc", "attn_q64.hip"), "-o", obj], check=True)
    objs = []
    for sub in ("csrc", "host"):
        d = os.path.join(PKG, "build", sub)
        objs += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".o") and not f.startswith("attn_q64")]
    out = os.path.join(VAR, f"libltxhip_{name}.so")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [obj, "-lz", "-ldl"], check=True)
    print("built", out)


def measure():
    sys.path.insert(0, PKG); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch, ltxhip
    from microbench import timeit
    heads = 32
    res = {}

    def t_us(Sq, Sk, iters=30):
        q, k, v = [(torch.randn(1, s, heads * 64, device="cuda") * 1.0).bfloat16() for s in (Sq, Sk, Sk)]
        qp = (q.float() * (0.125 * 1.4426950408889634)).bfloat16()
        fn = lambda: ltxhip.ops.attention_prescaled(qp, k, v, heads)
        timeit(fn, iters=5, warm=2)
        return min(timeit(fn, iters=iters, warm=2) for _ in range(3)) * 1e3

    timeit(lambda: None, iters=1, warm=0)
    us = t_us(4992, 4992, 40)
    res["c2_us"] = round(us, 1); res["c2_TF"] = round(4 * heads * 4992 * 4992 * 64 / us / 1e6, 1)
    for kind, env, rounds in (("big", None, 2), ("small", "0", 4)):
        ltxhip.set_option("attn_q64_big", env if env is not None else "16")
        a, b = t_us(4096, 1024), t_us(4096, 4096)
        it = (b - a) / (48 * rounds)
        res[f"{kind}_it"] = round(it, 4); res[f"{kind}_fix"] = round(a / rounds - 16 * it, 2)
        ltxhip.set_option("attn_q64_big", None)
    # in-kernel stamps of the last launch (diagnostic builds): cycles per key tile and the clock inside the loop
    import ctypes
    lib = ctypes.CDLL(os.environ.get("LTXHIP_LIB") or os.path.join(PKG, "libltxhip.so"))
    if hasattr(lib, "ltx_dbg_q64_stamps"):
        import numpy as np
        for kind, env in (("big", "16"), ("small", "0")):
            ltxhip.set_option("attn_q64_big", env)
            t_us(4096, 4096, 3)
            buf = (ctypes.c_ulonglong * (8 * 512))()
            lib.ltx_dbg_q64_stamps(buf, 8 * 512)
            a = np.array(buf[:], dtype=np.float64).reshape(-1, 8)
            a = a[a[:, 2] > 0]
            res[f"{kind}_cyc"] = round(float(np.median(a[:, 0] / a[:, 2])), 1)
            res[f"{kind}_GHz"] = round(float(np.median(a[:, 0] / a[:, 1]) * 0.1), 3)
            res[f"{kind}_pro"] = round(float(np.median(a[:, 4])), 0); res[f"{kind}_epi"] = round(float(np.median(a[:, 5])), 0)
            raw = np.array(buf[:], dtype=np.uint64).reshape(-1, 8); raw = raw[raw[:, 2] > 0]
            w6, w7 = raw[:, 6], raw[:, 7]
            parts = [w6 >> np.uint64(48), (w6 >> np.uint64(32)) & np.uint64(0xffff), (w6 >> np.uint64(16)) & np.uint64(0xffff), w6 & np.uint64(0xffff), w7 >> np.uint64(32), w7 & np.uint64(0xffffffff)]
            res[f"{kind}_pro_parts"] = [int(np.median(x)) for x in parts]   # C++ setup | issue Q+DMA | wait vmcnt | barrier | S0,max,sub,K1 | exp slice
        ltxhip.set_option("attn_q64_big", None)
    print(json.dumps(res), flush=True)


def run(names):
    libs = sorted(f for f in os.listdir(VAR) if f.startswith("libltxhip_") and f.endswith(".so"))
    if names: libs = [f"libltxhip_{n}.so" for n in names]
    for lib in libs:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "measure"], capture_output=True, text=True,
                           env=dict(os.environ, LTXHIP_LIB=os.path.join(VAR, lib)))
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(lib[len("libltxhip_"):-3], line[-1] if line else ("FAILED " + p.stderr[-400:]), flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "build": build(sys.argv[2], sys.argv[3:])
    elif cmd == "measure": measure()
    elif cmd == "run": run(sys.argv[2:])
