#!/bin/bash
# Runs on the GPU box: SQ counter passes (separate rocprofv3 --pmc runs, no tracing domains beside them) for bench.py's resident leg.
#   bash tools/profile_sq.sh c3      -> gpurun_out/prof/<workload>_sq{1,2,3}
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
w=${1:-c3}
k=1
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  rm -rf gpurun_out/prof/${w}_sq$k
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof/${w}_sq$k -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --legs resident > gpurun_out/prof/${w}_sq$k.log 2>&1 || echo "pass $k failed"
  k=$((k+1))
done
for k in 1 2 3; do python3 tools/pmc_table.py gpurun_out/prof/${w}_sq$k; done
