#!/usr/bin/env python3
"""Block-level timeline of conv_halo_kernel (the pipelined 128-wide tile) on the VAE's conv shapes, from in-kernel stamps
(-DHALO_TRACE build): per block entry -> first staging landed (prologue), the K loop (wall time on the 100 MHz clock and shader
cycles -> the clock held inside it), the wide epilogue, the store drain; plus how long a CU sits between two of its blocks.
  build  -> tools/variants/libltxhip_halotrace.so     run  (GPU box) one JSON line per shape"""
import json, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd"); VAR = os.path.join(ROOT, "tools", "variants"); HIPCC = "/opt/rocm/bin/hipcc"


def build():
    """build [NAME -D...]: extra defines make a second traced library (tools/variants/libltxhip_halotrace_NAME.so; LTXHIP_LIB selects it for `run`)"""
    os.makedirs(VAR, exist_ok=True)
    bdir = os.path.join(PKG, "build", "var"); os.makedirs(bdir, exist_ok=True)
    tag = "_" + sys.argv[2] if len(sys.argv) > 2 else ""
    obj = os.path.join(bdir, f"conv_halo_trace{tag}.o")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DHALO_TRACE=1"] + sys.argv[3:] + ["-x", "hip", "-c",
                    os.path.join(PKG, "csrc", "conv_halo.hip"), "-o", obj], check=True)
    objs = []
    for sub in ("csrc", "host"):
        d = os.path.join(PKG, "build", sub)
        objs += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".o") and not f.startswith("conv_halo")]
    out = os.path.join(VAR, f"libltxhip_halotrace{tag}.so")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [obj, "-lz", "-ldl"], check=True)
    print("built", out)


def run():
    os.environ.setdefault("LTXHIP_LIB", os.path.join(VAR, "libltxhip_halotrace.so"))
    sys.path.insert(0, PKG)
    import ctypes
    import numpy as np
    import torch
    import ltxhip
    ltxhip.lib.ltx_dbg_halo_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for name, C, T, H, W, resid in [("128ch conv1 (bias)", 128, 97, 128, 192, False), ("128ch conv2 (+resid)", 128, 97, 128, 192, True), ("256ch (+resid)", 256, 49, 64, 96, True), ("512ch (+resid)", 512, 25, 32, 48, True)]:
        x = torch.randn(1, T, H, W, C, device="cuda").bfloat16(); w = (torch.randn(C, C, 3, 3, 3, device="cuda") / math.sqrt(27 * C)).bfloat16(); b = torch.randn(C, device="cuda").bfloat16()
        r = torch.randn(1, T, H, W, C, device="cuda").bfloat16() if resid else None
        with ltxhip.options(gemm_plan="halo:128"):
            for _ in range(3): ltxhip.ops.conv3d(x, w, b, False, resid=r)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ltxhip.ops.conv3d(x, w, b, False, resid=r)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3          # (includes the op's weight repack: the stamps below are the kernel's own)
        nb = T * -(-H // 16) * -(-W // 16) * (C // 128)
        n = min(nb, 16384)
        buf = np.zeros(16384 * 8, dtype=np.uint64)
        assert ltxhip.lib.ltx_dbg_halo_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
        t = buf.reshape(16384, 8)[:n].astype(np.float64)
        us = lambda a, b_: float(np.median(t[:, a] - t[:, b_]) / 100)
        loop_us = us(2, 1); cyc = float(np.median(t[:, 6] - t[:, 5]))
        hw = buf.reshape(16384, 8)[:n, 7].astype(np.uint64)
        cu = (hw >> np.uint64(8)) & np.uint64(0xf) | (((hw >> np.uint64(12)) & np.uint64(0x3)) << np.uint64(4)) | (((hw >> np.uint64(13)) & np.uint64(0x7)) << np.uint64(6))
        gaps, last = [], {}
        order = np.argsort(t[:, 0])
        for bidx in order:
            key = (int(bidx) % 8, int(cu[bidx]))
            if key in last: gaps.append((t[bidx, 0] - last[key]) / 100)
            last[key] = t[bidx, 4]
        steps = 27 * (C // 64)
        rec = {"shape": name, "voxels": T * H * W, "blocks": nb, "stamped": n, "op_ms_with_repack": round(ms, 3),
               "kernel_span_us": round(float((t[:, 4].max() - t[:, 0].min()) / 100), 1),
               "block_us": round(us(4, 0), 2), "prologue_us": round(us(1, 0), 2), "loop_us": round(loop_us, 2), "us_per_step": round(loop_us / steps, 4),
               "epilogue_us": round(us(3, 2), 2), "store_drain_us": round(us(4, 3), 2),
               "loop_clock_GHz": round(cyc / (loop_us * 1e3), 3), "cycles_per_step": round(cyc / steps, 1), "mfma_cycles_per_step": 2 * 16 * 16 * 2,
               "cu_idle_between_blocks_us": {"median": round(float(np.median(gaps)), 2), "p90": round(float(np.percentile(gaps, 90)), 2), "n": len(gaps)} if gaps else None}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
