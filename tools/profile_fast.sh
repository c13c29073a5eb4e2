#!/bin/bash
# Runs on the GPU box: SQ / TA / TCP counters of the probe kernels on bench.py's resident leg (one rocprofv3 --pmc run per set).
#   bash tools/profile_fast.sh c3      -> gpurun_out/prof/<workload>_pf{1..}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
w=${1:-c3}
k=1
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "TA_TA_BUSY_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf gpurun_out/prof/${w}_pf$k
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof/${w}_pf$k -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --legs resident > gpurun_out/prof/${w}_pf$k.log 2>&1 || echo "pass $k failed: $set"
  python3 tools/pmc_table.py gpurun_out/prof/${w}_pf$k 2>/dev/null | grep "k_probe"
  k=$((k+1))
done
