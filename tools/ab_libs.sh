#!/bin/bash
# Runs on the GPU box: the probe phase of several builds of the library on ONE box, interleaved (boxes differ by a few per cent).
#   AB_LIBS="tools/lib_vA.so tools/lib_vB.so" bash tools/ab_libs.sh c3 c2 c4
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for w in "$@"; do
    for ent in $AB_LIBS; do            # "path/to/lib.so" or "path/to/lib.so:VAR=1" (an environment switch of that build)
      lib=${ent%%:*}; var=${ent#*:}; [ "$var" = "$ent" ] && var=MM_AB_NONE=1; var=${var//,/ }   # several switches: A=1,B=1
      env $var MMANNOT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --workload $w --steps 12 --warmup 2 --no-cpu-baseline --legs resident 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ph = d['phase_ms_per_step']
print('rep $rep  $w  %-44s probe %.1f us   kernel_only %.2f G/s   passed on %.4f  path %s  sort %.1f merge %.1f us' % ('$ent', d['roofline']['avg_launch_ms'] * 1e3, d['kernel_only']['value'] / 1e9, d['roofline'].get('passed_on_frac', -1), d['roofline'].get('probe_path', '?'), ph['sort'] * 1e3, ph['merge'] * 1e3))"
    done
  done
done
