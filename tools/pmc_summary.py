#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel (substring filter). usage: pmc_summary.py <dir> <substr> [<substr>...]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; subs = sys.argv[2:]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for s in subs:
        if s in n:
            key = s + " grid=" + r["Grid_Size"] + " wg=" + r["Workgroup_Size"] + " vgpr=" + r["VGPR_Count"] + " lds=" + r["LDS_Block_Size"]
            a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in acc.items():
    print(k)
    for c, (s, n) in sorted(v.items()):
        print(f"   {c:32s} avg={s/n:14.1f}  n={n}")
