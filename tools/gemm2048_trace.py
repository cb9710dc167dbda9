#!/usr/bin/env python3
"""Where the time of the K = N = 2048 linear layers (to_out / q2 / out2: 588 launches, a quarter of the dominant kernel's time, 0.39-0.42
of the MFMA peak) goes: gemm_asm16_kernel<160, 256> built with in-kernel stamps (generator trace=1 + -DGEMM_ASM_TRACE), one launch
per epilogue flavour, per-block timeline on the 100 MHz clock.
  build            -> tools/variants/libltxhip_g2048trace.so     (CPU box)
  run              (GPU box) prints one JSON line per flavour; LTXHIP_LIB selects the variant library"""
import json, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "candle-video_amd")


def build():
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_asm_tune.py"), "build", "g2048trace", "trace=1"], check=True)


def run():
    os.environ.setdefault("LTXHIP_LIB", os.path.join(ROOT, "tools", "variants", "libltxhip_g2048trace.so"))
    sys.path.insert(0, PKG)
    import ctypes
    import numpy as np
    import torch
    import ltxhip
    M, N = 4992, 2048
    dev = "cuda"
    for K in (2048, 8192):
        x = torch.randn(M, K, device=dev).bfloat16()
        ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(6)]          # rotate weights: cold-ish like the pipeline
        b = torch.randn(N, device=dev).bfloat16(); h = torch.randn(M, N, device=dev).bfloat16(); gate = torch.randn(1, 6 * N, device=dev)
        flavours = {"bias (q2)": lambda w: ltxhip.ops.linear(x, w, b),
                    "bias+rowsq (q2 as shipped)": lambda w: ltxhip.ops.linear_rowsq(x, w, b),
                    "resid+rowsq (out2)": lambda w: ltxhip.ops.linear_rowsq(x, w, b, epi=3, resid=h),
                    "gate_resid (to_out)": lambda w: ltxhip.ops.linear(x, w, b, epi=2, resid=h, gate=gate, rows_per_batch=M)}
        for tag, fn in flavours.items():
            for i in range(12): fn(ws[i % 6])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(30): fn(ws[i % 6])
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 30
            fn(ws[0]); torch.cuda.synchronize()
            nb = -(-M // 160) * (N // 256)
            buf = np.zeros(1024 * 4 * 16, dtype=np.uint32)
            assert ltxhip.lib.ltx_dbg_gemm_asm16_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
            t = buf.reshape(1024, 4, 16)[:nb].astype(np.float64)
            w0 = t[:, 0, :]
            first = w0[:, 8].min()
            d = lambda a, bb: round(float(((w0[:, a] - w0[:, bb]) % 2**32).mean() / 100), 2)
            rec = {"K": K, "flavour": tag, "plan": ltxhip.ops.gemm_plan(M, N, K), "us_per_launch_stream": round(us, 1), "TFLOPs": round(2 * M * N * K / us / 1e6),
                   "blocks": nb, "span_us": round(float((w0[:, 11].max() - first) / 100), 2),
                   "entry_spread_us": round(float((w0[:, 8].max() - first) / 100), 2),
                   "entry_to_loop_asm": d(9, 8), "prologue_first_two_ksteps_landed": round(float(((w0[:, 7] - w0[:, 9]) % 2**32).mean() / 100), 2),
                   "loop_total": d(10, 9), "epi_wait_bar0": d(13, 10), "epi_acc_to_lds": d(14, 13), "epi_bar1": d(15, 14), "epi_rows_pass0_and_all_of_pass1": d(11, 15),
                   "exit_spread_us": round(float((w0[:, 11].max() - w0[:, 11].min()) / 100), 2),
                   "cyc_per_kstep": round(float(t[:, :, :7].sum(axis=2).mean() / (K // 64 - 1)), 1)}
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
