#!/bin/bash
# Runs on the GPU box: more SQ / SQC counter passes for bench.py's resident leg (issue mix, instruction cache), one rocprofv3 --pmc run each.
#   bash tools/profile_sq2.sh c3      -> gpurun_out/prof/<workload>_sx{1..}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
w=${1:-c3}
k=1
for set in "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  rm -rf gpurun_out/prof/${w}_sx$k
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof/${w}_sx$k -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --legs resident > gpurun_out/prof/${w}_sx$k.log 2>&1 || echo "pass $k failed: $set"
  python3 tools/pmc_table.py gpurun_out/prof/${w}_sx$k 2>/dev/null | grep k_probe
  k=$((k+1))
done
