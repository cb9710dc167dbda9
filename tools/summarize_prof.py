#!/usr/bin/env python3
"""Condense rocprofv3 CSVs (kernel trace + optional PMC passes) into a small tracked summary under profiles/.

Only the LAST video period of the profiled run is summarised: periods are delimited by the denorm_mix launch that opens a
VAE decode (decode of video n-1 + denoise loop of video n = one video's worth of launches), so warm-up work (first-call
GEMM plan measurement, workspace allocation) never enters the numbers.

usage: summarize_prof.py <stats_dir> <out_prefix> [--pmc FETCH_SIZE=<dir> --pmc WRITE_SIZE=<dir>] [--sq <dir>]
  --sq <dir>: a pass with SQ / GRBM counters (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY,
  SQ_WAIT_ANY, GRBM_GUI_ACTIVE ...): per kernel class, the per-launch average of every counter and the derived MFMA utilisation
  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), the clock = GRBM_GUI_ACTIVE / 8 / launch duration."""
import csv, glob, json, os, re, sys
from collections import defaultdict

DELIM = re.compile(r"denorm_mix_kernel")      # launched once per video (start of the VAE decode): one period = one video


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:100]


def targs(n):
    m = re.search(r"<(.*)>\(", n)
    return [a.strip() for a in m.group(1).split(",")] if m else []


def classify(n):
    a = targs(n)
    if "gemm_big_kernel" in n:
        return "conv3d implicit GEMM" if len(a) > 5 and a[5] == "true" else "linear GEMM"
    if "gemm_asm16_conv_kernel" in n:
        return "conv3d implicit GEMM"
    if "gemm_ring_kernel" in n:                  # <BM, BN, NS, EPI, SPEC, CONV>
        return "conv3d implicit GEMM" if len(a) > 5 and a[5] == "true" else "linear GEMM"
    if "gemm_asm16_kernel" in n or "gemm_asm_kernel" in n:
        return "linear GEMM"
    if "conv_halo_kernel" in n:
        return "conv3d implicit GEMM"
    if "gemm_p8_kernel" in n:
        return "conv3d implicit GEMM" if len(a) > 4 and a[4] == "true" else "linear GEMM"
    if "gemm_kernel" in n:
        # rocprofv3 cannot demangle the bf16 template argument of this kernel: "..., ELb1E>" is the CONV=true form
        return "conv3d implicit GEMM" if "Lb1" in n else "linear GEMM"
    if "attn_cross64_kernel" in n:
        return "attention (cross / generic)"
    if "attn_pipe64_kernel" in n or "attn_q64_kernel" in n or "attn_q128_kernel" in n:
        return "attention (self, q-prescaled)"
    if "attn_bf16_kernel" in n:
        return "attention (self, q-prescaled)" if len(a) > 1 and a[1] == "true" else "attention (cross / generic)"
    for k in ("attn_f32", "rownorm", "qknorm_rope", "cast_kernel", "pack_conv", "guidance", "denorm", "rope_table"):
        if k in n:
            return k
    return "other (torch RNG/fill for synthetic inputs, misc)"


def last_video(rows, name_key):
    idx = [i for i, r in enumerate(rows) if DELIM.search(r[name_key])]
    if len(idx) >= 2:
        return rows[idx[-2] + 1: idx[-1] + 1], True
    return rows, False


def main():
    stats_dir, out = sys.argv[1], sys.argv[2]
    pmc = {}
    a = sys.argv[3:]
    sq_dir = None
    for i, x in enumerate(a):
        if x == "--pmc":
            k, d = a[i + 1].split("="); pmc[k] = d
        if x == "--sq":
            sq_dir = a[i + 1]
    f = glob.glob(os.path.join(stats_dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    win, ok = last_video(rows, "Kernel_Name")
    per = defaultdict(lambda: [0.0, 0]); cls = defaultdict(lambda: [0.0, 0])
    for r in win:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        per[r["Kernel_Name"]][0] += d; per[r["Kernel_Name"]][1] += 1
        c = classify(r["Kernel_Name"]); cls[c][0] += d; cls[c][1] += 1
    tot = sum(v[0] for v in per.values())
    span = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:28]:
        lines.append(f"| `{short(n)}` | {c} | {t/1e6:.2f} | {t/1e3/c:.1f} | {100*t/tot:.1f} |")
    summary = {"window": "last video period of the run" if ok else "whole run (no video delimiter found)", "kernel_busy_ms": tot / 1e6, "wall_span_ms": span / 1e6,
               "launches": len(win),
               "classes": {k: {"ms_per_video": v[0] / 1e6, "launches_per_video": v[1], "avg_us": v[0] / 1e3 / max(v[1], 1)} for k, v in sorted(cls.items(), key=lambda kv: -kv[1][0])}}
    # the VAE decode launch by launch (conv class, launch order): duration, workgroups, rounds of the 256 CUs at one block per CU
    dec = []
    for r in win:
        if classify(r["Kernel_Name"]) != "conv3d implicit GEMM":
            continue
        wg = None
        try:
            gx, wx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)))
            wg = gx // wx if wx else None
        except (TypeError, ValueError):
            pass
        dec.append({"kernel": short(r["Kernel_Name"])[:60], "us": round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1),
                    "workgroups": wg, "rounds_of_256": round(wg / 256.0, 2) if wg else None})
    summary["decode_launches"] = dec
    for name, d in pmc.items():
        fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            continue
        prow = [r for r in csv.DictReader(open(fs[0])) if r.get("Counter_Name") == name]
        prow.sort(key=lambda r: int(r["Dispatch_Id"]))
        pwin, _ = last_video(prow, "Kernel_Name")
        acc = defaultdict(lambda: [0.0, 0])
        for r in pwin:
            c = classify(r["Kernel_Name"]); acc[c][0] += float(r["Counter_Value"]); acc[c][1] += 1
        # FETCH_SIZE / WRITE_SIZE are reported in KiB (MI355X_MICROARCH.md, HBM/rocprofv3 section); gfx950: double FETCH_SIZE
        summary.setdefault("pmc", {})[name] = {k: {"per_launch_MB_raw": v[0] * 1024 / max(v[1], 1) / 1e6, "launches": v[1]} for k, v in acc.items()}
    if sq_dir:
        fs = glob.glob(os.path.join(sq_dir, "**", "*counter_collection.csv"), recursive=True)
        if fs:
            rows_sq = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Dispatch_Id"]))
            # one row per (dispatch, counter): window by dispatch order using the delimiter kernel
            disp = {}
            for r in rows_sq:
                disp.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
                if "Start_Timestamp" in r and r.get("Start_Timestamp"):
                    disp[int(r["Dispatch_Id"])]["dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            dl = [dict(v, Dispatch_Id=k) for k, v in sorted(disp.items())]
            win_sq, _ = last_video(dl, "Kernel_Name")
            acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
            for r in win_sq:
                c = classify(r["Kernel_Name"]); cnt[c] += 1
                for k, v in r.items():
                    if k not in ("Kernel_Name", "Dispatch_Id"):
                        acc[c][k] += v
            sq = {}
            for c in acc:
                d = {k: v / cnt[c] for k, v in acc[c].items()}
                d["launches"] = cnt[c]
                if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE", 0) > 0:
                    d["mfma_busy_frac_of_real_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                if d.get("dur_ns", 0) > 0 and d.get("GRBM_GUI_ACTIVE", 0) > 0:
                    d["clock_GHz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / d["dur_ns"]
                    if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
                        d["mfma_busy_frac_of_2.4GHz_peak"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (d["dur_ns"] * 2.4)
                sq[c] = d
            summary["sq"] = sq
    open(out + ".json", "w").write(json.dumps(summary, indent=1))
    md = (f"# rocprofv3 --kernel-trace summary ({os.path.basename(out)}) — {summary['window']}\n\n"
          f"kernel busy time {tot/1e6:.1f} ms over a wall span of {span/1e6:.1f} ms, {len(win)} launches\n\n" + "\n".join(lines) +
          "\n\n## by class (per video)\n\n" +
          "\n".join(f"- {k}: {v['ms_per_video']:.2f} ms, {v['launches_per_video']} launches, avg {v['avg_us']:.1f} us" for k, v in summary["classes"].items()) + "\n")
    if summary.get("decode_launches"):
        md += "\n## VAE decode, launch by launch (conv class)\n\n| # | kernel | us | workgroups | rounds of 256 CUs |\n|---|---|---|---|---|\n"
        for i, r in enumerate(summary["decode_launches"]):
            md += f"| {i} | `{r['kernel']}` | {r['us']} | {r['workgroups']} | {r['rounds_of_256']} |\n"
    if "pmc" in summary:
        md += "\n## PMC (raw counter value x 1024 B, per launch; FETCH_SIZE must be doubled on gfx950)\n\n"
        for name, d in summary["pmc"].items():
            md += f"### {name}\n\n" + "\n".join(f"- {k}: {v['per_launch_MB_raw']:.1f} MB/launch over {v['launches']} launches" for k, v in sorted(d.items(), key=lambda kv: -kv[1]['per_launch_MB_raw'])) + "\n\n"
    if "sq" in summary:
        md += "\n## SQ / GRBM counters per launch (separate pass) and MFMA utilisation\n\n"
        for c, d in sorted(summary["sq"].items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * kv[1]["launches"]):
            if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
                continue
            md += (f"- {c} ({d['launches']} launches): MFMA busy {100 * d.get('mfma_busy_frac_of_real_cycles', 0):.1f} % of real SIMD cycles"
                   + (f", {100 * d['mfma_busy_frac_of_2.4GHz_peak']:.1f} % of the 2.4 GHz peak, clock {d['clock_GHz']:.2f} GHz" if "clock_GHz" in d else "")
                   + "; " + ", ".join(f"{k} {v:.3g}" for k, v in sorted(d.items()) if k.startswith(("SQ_", "GRBM_"))) + "\n")
    open(out + ".md", "w").write(md)
    print(md)


if __name__ == "__main__":
    main()
