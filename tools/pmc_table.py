#!/usr/bin/env python3
"""Print per-kernel means (in millions) of every counter in a rocprofv3 --pmc output directory."""
import collections, csv, glob, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    if k.startswith("k_"):
        print(k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(d[k].items())}, "launches", len(next(iter(d[k].values()))))
