#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace run (the newest *kernel_trace.csv under the directory given):
per kernel name, mean duration and mean idle time before it, over the last `n` dispatches.   python tools/gaps.py DIR [n]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
f = max(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
agg = defaultdict(lambda: [0, 0.0, 0.0]); prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:32] + " scratch=" + r["Scratch_Size"]
    a = agg[k]; a[0] += 1; a[1] += (e - s) / 1e3; a[2] += ((s - prev) / 1e3 if prev else 0.0); prev = e
tot_d = sum(a[1] for a in agg.values()); tot_g = sum(a[2] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-46s calls %4d  dur %8.1f us  idle before %6.1f us" % (k, a[0], a[1] / a[0], a[2] / a[0]))
print("busy %.1f us  idle %.1f us  (idle share %.1f %%)" % (tot_d, tot_g, 100 * tot_g / (tot_d + tot_g)))
