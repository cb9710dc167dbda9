#!/usr/bin/env python3
"""Variant builder / runner for the generated K loop of csrc/gemm_asm.hip (same scheme as tools/attn_q64_tune.py).
  build NAME [key=value ...]   -> tools/variants/libltxhip_NAME.so
  run [NAME ...]               (GPU box) TF/s of each variant on the square and DiT shapes, forced tile asm256x256 unless TILE=..."""
import json, math, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd"); VAR = os.path.join(ROOT, "tools", "variants"); HIPCC = "/opt/rocm/bin/hipcc"


def build(name, opts):
    os.makedirs(VAR, exist_ok=True)
    defs = [o for o in opts if o.startswith("-D")]         # compile-time switches of gemm_asm.hip; the rest are generator options
    opts = [o for o in opts if not o.startswith("-D")]
    bdir = os.path.join(PKG, "build", "var"); os.makedirs(bdir, exist_ok=True)
    inc = os.path.join(bdir, f"gemm_loop_{name}.inc")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemm_asm.py"), "--out", inc] + opts, check=True)
    obj = os.path.join(bdir, f"gemm_asm_{name}.o")
    extra = (["-DGEMM_ASM_REG=1"] if "stage=reg" in opts else []) + (["-DGEMM_ASM_TRACE=1"] if "trace=1" in opts else []) + defs
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", f'-DGEMM_ASM_LOOP_INC="{inc}"'] + extra + ["-x", "hip", "-c",
                    os.path.join(PKG, "csrc", "gemm_asm.hip"), "-o", obj], check=True)
    objs = []
    for sub in ("csrc", "host"):
        d = os.path.join(PKG, "build", sub)
        objs += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".o") and not f.startswith("gemm_asm")]
    out = os.path.join(VAR, f"libltxhip_{name}.so")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [obj, "-lz", "-ldl"], check=True)
    print("built", out)


def measure():
    sys.path.insert(0, PKG); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch, ltxhip
    from microbench import timeit
    res = {}
    tile = os.environ.get("TILE", "asm256x256")
    os.environ["LTX_GEMM_ASM_TILE"] = tile
    mode = os.environ.get("ASM_MODE", "1")                 # "1": 32x32x16 loop, "16": 16x16x32 loop
    os.environ["LTX_GEMM_ASM"] = "0" if tile == "big" else mode
    zero = os.environ.get("ZERO", "0") == "1"              # all-zero operands: no data toggling, the clock stays up - cycles, not power
    for name, M, N, K in [("sq8192", 8192, 8192, 8192), ("sq4096", 4096, 4096, 4096), ("qkv", 4992, 6144, 2048), ("ff2", 4992, 2048, 8192), ("k16k", 4096, 4096, 16384)]:
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
        if zero: x.zero_(); w.zero_()
        fn = lambda: ltxhip.ops.linear(x, w, None)
        t = min(timeit(fn, iters=10, warm=2) for _ in range(3))
        res[name] = round(2 * M * N * K / t / 1e9, 1)
    if tile != "big":                      # bit-identity against gemm_big on a ragged shape (M, N not multiples of the tile, short K)
        x = torch.randn(3001, 320, device="cuda").bfloat16(); w = (torch.randn(4104, 320, device="cuda") / 18).bfloat16(); b = torch.randn(4104, device="cuda").bfloat16()
        y = ltxhip.ops.linear(x, w, b)
        os.environ["LTX_GEMM_ASM"] = "0"; ref = ltxhip.ops.linear(x, w, b); os.environ["LTX_GEMM_ASM"] = mode
        res["equal_big"] = bool(torch.equal(y.view(torch.int16), ref.view(torch.int16))); res["max_diff"] = float((y.float() - ref.float()).abs().max())
    if hasattr(ltxhip.lib, "ltx_dbg_gemm_asm16_trace") and tile != "big":
        # mean cycles per K-step of each segment over all waves, and the per-block timeline on the 100 MHz clock
        import ctypes, numpy as np
        names = ["A-B reads", "B-C bar1", "C-D dma", "D-E half", "E-G bar2", "G-H reads", "H-A top"]
        for tag, m, n, k in (("k16k", 4096, 4096, 16384), ("sq8192", 8192, 8192, 8192), ("qkv", 4992, 6144, 2048), ("ff2", 4992, 2048, 8192)):
            x = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).bfloat16()
            if zero: x.zero_(); w.zero_()
            for _ in range(3): ltxhip.ops.linear(x, w, None)
            torch.cuda.synchronize()
            nb = -(-m // 256) * -(-n // 256)
            buf = np.zeros(1024 * 4 * 16, dtype=np.uint32)
            assert ltxhip.lib.ltx_dbg_gemm_asm16_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
            t = buf.reshape(1024, 4, 16)[:nb].astype(np.float64)
            steps = k // 64 - 1
            res[tag + "_trace"] = {nm: round(float(t[:, :, i].mean() / steps), 1) for i, nm in enumerate(names)}
            res[tag + "_cyc_per_step"] = round(float(t[:, :, :7].sum(axis=2).mean() / steps), 1)
            w0 = t[:, 0, :]
            res[tag + "_block_us"] = {"entry_to_loop": round(float((w0[:, 9] - w0[:, 8]).mean() / 100), 2), "prologue": round(float(((w0[:, 7] - w0[:, 9]) % 2**32).mean() / 100), 2),
                                      "loop": round(float((w0[:, 10] - w0[:, 9]).mean() / 100), 2),
                                      "epilogue": round(float((w0[:, 11] - w0[:, 10]).mean() / 100), 2),
                                      "epi_bar0": round(float((w0[:, 13] - w0[:, 10]).mean() / 100), 2), "epi_cvt": round(float((w0[:, 14] - w0[:, 13]).mean() / 100), 2),
                                      "epi_bar1": round(float((w0[:, 15] - w0[:, 14]).mean() / 100), 2), "epi_copy": round(float((w0[:, 11] - w0[:, 15]).mean() / 100), 2)}
            hw = buf.reshape(1024, 4, 16)[:nb, 0, 12]
            cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x3) << 4) | (((hw >> 13) & 0x7) << 6)      # cu_id, sh_id, se_id fields; XCC from the block's order
            # gap between consecutive blocks that ran on the same (xcc = bid % 8, cu): exit of one to entry of the next
            gaps = []
            key = {}
            for b in np.argsort(w0[:, 8]):
                kk = (b % 8, int(cu[b]))
                if kk in key: gaps.append((w0[b, 8] - key[kk]) / 100)
                key[kk] = w0[b, 11]
            if gaps: res[tag + "_gap_us"] = {"mean": round(float(np.mean(gaps)), 2), "p90": round(float(np.percentile(gaps, 90)), 2), "n": len(gaps)}
            res[tag + "_span_us"] = round(float((w0[:, 11].max() - w0[:, 8].min()) / 100), 1)
    # cycles per K-step from the slope over K at fixed 4096 x 4096 (256 tiles = one per CU), assuming 2.0 GHz
    res["us_per_kstep"] = round((2 * 4096 * 4096 * 16384 / res["k16k"] / 1e6 - 2 * 4096 ** 3 / res["sq4096"] / 1e6) / (256 - 64), 4)
    print(json.dumps(res), flush=True)


def run(names):
    libs = sorted(f for f in os.listdir(VAR) if f.startswith("libltxhip_") and f.endswith(".so"))
    if names: libs = [f"libltxhip_{n}.so" for n in names]
    for lib in libs:
        shutil.copyfile(os.path.join(VAR, lib), os.path.join(PKG, "libltxhip.so"))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "measure"], capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(lib[len("libltxhip_"):-3], line[-1] if line else ("FAILED " + p.stderr[-300:]), flush=True)


if __name__ == "__main__":
    {"build": lambda: build(sys.argv[2], sys.argv[3:]), "measure": measure, "run": lambda: run(sys.argv[2:])}[sys.argv[1]]()
