#!/usr/bin/env python3
"""Idle gaps in a rocprofv3 kernel trace (kernel_trace.csv): total idle time inside the last third of the run, and the kernels that
precede / follow the largest gaps.  usage: gap_report.py <rocprof output dir> [min_gap_us]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) * 2 // 3:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"launches {len(rows)}, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms")
acc = collections.defaultdict(lambda: [0, 0.0])
tot_small = 0.0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g >= thr:
        k = (a["Kernel_Name"][:60], b["Kernel_Name"][:60])
        acc[k][0] += 1; acc[k][1] += g
    elif g > 0:
        tot_small += g
print(f"gaps under {thr / 1e3:.0f} us: {tot_small / 1e6:.1f} ms in total")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{v[1] / 1e6:8.2f} ms in {v[0]:5d} gaps (avg {v[1] / v[0] / 1e3:7.1f} us)  after {k[0]}  ->  before {k[1]}")
