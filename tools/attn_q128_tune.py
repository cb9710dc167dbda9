#!/usr/bin/env python3
"""Variant builder / runner for the generated loop of csrc/attn_q128.hip (same scheme as tools/gemm_asm_tune.py).
  build NAME [key=value ...]   -> tools/variants/libltxhip_NAME.so      run [NAME ...]   (GPU box) TF/s at S = 17556 and 4992"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd"); VAR = os.path.join(ROOT, "tools", "variants"); HIPCC = "/opt/rocm/bin/hipcc"


def build(name, opts):
    os.makedirs(VAR, exist_ok=True)
    bdir = os.path.join(PKG, "build", "var"); os.makedirs(bdir, exist_ok=True)
    inc = os.path.join(bdir, f"q128_loop_{name}.inc")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_q128_asm.py"), "--out", inc] + opts, check=True)
    obj = os.path.join(bdir, f"attn_q128_{name}.o")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", f'-DQ128_LOOP_INC="{inc}"'] + (["-DQ128_NO_FALLBACK=1"] if any(o.startswith("abl=") for o in opts) else []) + ["-x", "hip", "-c",
                    os.path.join(PKG, "csrc", "attn_q128.hip"), "-o", obj], check=True)
    objs = []
    for sub in ("csrc", "host"):
        d = os.path.join(PKG, "build", sub)
        objs += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".o") and not f.startswith("attn_q128")]
    out = os.path.join(VAR, f"libltxhip_{name}.so")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [obj, "-lz", "-ldl"], check=True)
    print("built", out)


def run(names):
    libs = sorted(f for f in os.listdir(VAR) if f.startswith("libltxhip_") and f.endswith(".so"))
    if names: libs = [f"libltxhip_{n}.so" for n in names]
    for lib in libs:
        shutil.copyfile(os.path.join(VAR, lib), os.path.join(PKG, "libltxhip.so"))
        res = {}
        for S in (17556, 4992):
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn128_time.py")], capture_output=True, text=True, env=dict(os.environ, S=str(S)))
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1]) if line else {"TF": None, "err": p.stderr[-200:]}
            res[S] = (d.get("TF"), d.get("rel_l2_vs_f32"))
        print(lib[len("libltxhip_"):-3], res, flush=True)


if __name__ == "__main__":
    {"build": lambda: build(sys.argv[2], sys.argv[3:]), "run": lambda: run(sys.argv[2:])}[sys.argv[1]]()
