#!/usr/bin/env python3
"""Condense rocprofv3 CSVs (kernel stats + optional PMC passes) into a small tracked summary under profiles/.
usage: summarize_prof.py <stats_dir> <out_prefix> [--pmc FETCH_SIZE=<dir> --pmc WRITE_SIZE=<dir>] [--videos N]"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)([a-z_0-9]+?)I(.*)E+v", n)
    if n.startswith("_ZN12_GLOBAL__N_1"):
        m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9A-Z]+?)I(.*?)EEv", n)
        if m:
            return f"{m.group(1)}<{m.group(2)}>"
    return n[:90]


def classify(n):
    if "gemm_big_kernel" in n or "gemm_kernel" in n:
        conv = "Lb1" in n or ", true>" in n
        return "conv3d implicit GEMM" if conv else "linear GEMM"
    for k in ("attn_bf16", "attn_f32", "rownorm", "qknorm_rope", "cast_kernel", "pack_conv", "guidance", "denorm", "rope_table"):
        if k in n:
            return k
    return "other (torch RNG/fill for synthetic inputs, misc)"


def main():
    stats_dir, out = sys.argv[1], sys.argv[2]
    pmc = {}; videos = 3
    a = sys.argv[3:]
    for i, x in enumerate(a):
        if x == "--pmc": k, d = a[i + 1].split("="); pmc[k] = d
        if x == "--videos": videos = int(a[i + 1])
    f = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    cls = defaultdict(lambda: [0.0, 0])
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:24]:
        lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
    for r in rows:
        c = classify(r["Name"]); cls[c][0] += float(r["TotalDurationNs"]); cls[c][1] += int(r["Calls"])
    summary = {"total_gpu_ms": tot / 1e6, "videos": videos, "gpu_ms_per_video": tot / 1e6 / videos,
               "classes": {k: {"ms_per_video": v[0] / 1e6 / videos, "launches_per_video": v[1] / videos, "avg_us": v[0] / 1e3 / max(v[1], 1)} for k, v in sorted(cls.items(), key=lambda kv: -kv[1][0])}}
    for name, d in pmc.items():
        fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(fs[0])):
            if r.get("Counter_Name") != name:
                continue
            c = classify(r["Kernel_Name"]); acc[c][0] += float(r["Counter_Value"]); acc[c][1] += 1
        # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B? (derived: *_sum*64/1024) -> bytes = value*1024
        summary.setdefault("pmc", {})[name] = {k: {"per_launch_MB": v[0] * 1024 / max(v[1], 1) / 1e6, "launches": v[1]} for k, v in acc.items()}
    open(out + ".json", "w").write(json.dumps(summary, indent=1))
    open(out + ".md", "w").write(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(out)})\n\nGPU time {tot/1e6:.1f} ms over {videos} videos "
                                 f"({tot/1e6/videos:.1f} ms/video)\n\n" + "\n".join(lines) + "\n\n## by class (per video)\n\n" +
                                 "\n".join(f"- {k}: {v['ms_per_video']:.2f} ms, {v['launches_per_video']:.0f} launches, avg {v['avg_us']:.1f} us" for k, v in summary["classes"].items()) + "\n")
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
