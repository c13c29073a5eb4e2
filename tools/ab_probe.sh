#!/bin/bash
# Runs on the GPU box: A/B of the probe variants on bench.py's resident leg, with per-kernel times from rocprofv3 --kernel-trace --stats.
#   bash tools/ab_probe.sh TAG w1 w2 ...   -> gpurun_out/ab_<TAG>_<workload>{,_old}.{json,csv}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
for w in "$@"; do
  for v in ${AB_VARIANTS:-new old}; do
    unset MMANNOT_AMD_NO_FAST MMANNOT_AMD_FAST_SHIFT
    if [ $v = old ]; then export MMANNOT_AMD_NO_FAST=1; fi
    case $v in s*) export MMANNOT_AMD_FAST_SHIFT=${v#s};; esac      # s10: the fast table at 1024-base bins
    rm -rf gpurun_out/prof/ab_${tag}_${w}_$v
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ab_${tag}_${w}_$v -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --legs resident > gpurun_out/ab_${tag}_${w}_$v.json 2> gpurun_out/ab_${tag}_${w}_$v.err
    f=$(ls gpurun_out/prof/ab_${tag}_${w}_$v/*/*kernel_stats.csv 2>/dev/null | head -1)
    echo "== $w $v"; python3 - "$f" gpurun_out/ab_${tag}_${w}_$v.json <<'P'
import csv, sys, json
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("  probe phase %.1f us  passed on %.4f  fast slots %.0f MB  kernel_only %.2f G/s" % (d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["passed_on_frac"], d["roofline"]["fast_slot_bytes"] / 1e6, d["kernel_only"]["value"] / 1e9))
except Exception as e:
    print("  (no json:", e, ")")
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("k_") and not n.startswith("k_synth"):
        print("  %-28s calls %4s  avg %9.1f us  total %6.2f %%" % (n[:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
P
  done
done
