set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c3 c2 c4 c5; do
  timeout -k 10 500 python3 bench.py --workload $w > gpurun_out/r3_bench_${w}_default.json 2> gpurun_out/r3_bench_${w}_default.err || exit 1
done
bash tools/profile_all.sh c3 c2 c4 c5 || exit 1
bash tools/profile_fast.sh c3 > gpurun_out/r3_profile_fast_c3.txt 2>&1
# scale sensitivity (round-2 review item 7)
for args in "--annotation-scale 1.5" "--annotation-scale 3.5" "--annotation-scale 1.5 --scaffolds 2000" "--annotation-scale 3.5 --scaffolds 2000"; do
  tag=$(echo "$args" | tr -d ' -' | tr '.' 'p')
  timeout -k 10 300 python3 bench.py --workload c3 --steps 12 --warmup 5 --no-cpu-baseline --legs resident $args > gpurun_out/r3_scale_$tag.json 2> gpurun_out/r3_scale_$tag.err || exit 1
  for v in MMANNOT_AMD_FAST=1 MMANNOT_AMD_NO_FAST=1; do
    env $v timeout -k 10 300 python3 bench.py --workload c3 --steps 12 --warmup 5 --no-cpu-baseline --legs resident $args > gpurun_out/r3_scale_${tag}_$v.json 2>> gpurun_out/r3_scale_$tag.err || exit 1
  done
done
for tag in big scaf; do
  if [ $tag = big ]; then a="--annotation-scale 3.5"; else a="--annotation-scale 1.5 --scaffolds 2000"; fi
  rm -rf gpurun_out/prof/c3${tag}_trace
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/c3${tag}_trace -- python3 bench.py --workload c3 --steps 10 --warmup 5 --no-cpu-baseline --legs resident $a > gpurun_out/prof/c3${tag}_trace.log 2>&1
done
python3 - <<'P' > gpurun_out/r3_scale.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_scale_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    ib = d["config"]["index_bytes"]; r = d["roofline"]
    print(f.split("r3_scale_")[1], "| intervals %d | chromosomes %s | slots %.0f MB (shift %d) flat %.0f MB full %.0f MB fast slots %.0f MB | probe %.1f us (%s) | kernel_only %.1f G/s" % (
        d["config"]["intervals"], d["config"].get("chromosomes", "?"), ib["slot_bytes"] / 1e6, ib["slot_shift"], ib["flat_bytes"] / 1e6, (ib["interval_bytes"] + ib["bin_bytes"] + ib["active_bytes"]) / 1e6,
        r["fast_slot_bytes"] / 1e6, r["avg_launch_ms"] * 1e3, r["probe_path"], d["kernel_only"]["value"] / 1e9))
P
cat gpurun_out/r3_scale.txt
