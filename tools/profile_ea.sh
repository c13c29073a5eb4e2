#!/bin/bash
# Runs on the GPU box: the L2's memory-side read requests by size (32 / 64 / 128 B) and destination, for bench.py's resident leg -
# the exact bytes behind FETCH_SIZE, which tallies every request at 64 B.  One rocprofv3 --pmc run per set, nothing else traced.
#   bash tools/profile_ea.sh c3      -> gpurun_out/prof/<workload>_ea{1,2}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
w=${1:-c3}
k=1
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum"; do
  rm -rf gpurun_out/prof/${w}_ea$k
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof/${w}_ea$k -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --legs resident > gpurun_out/prof/${w}_ea$k.log 2>&1 || echo "pass $k failed: $set"
  python3 tools/pmc_table.py gpurun_out/prof/${w}_ea$k 2>/dev/null | grep "k_probe\|k_rs_scatter\|k_leaf\|k_synth"
  k=$((k+1))
done
