#!/bin/bash
# Runs on the GPU box: texture-addresser / vector-L1 counters of bench.py's resident leg (is k_probe bound by the per-CU address path?).
#   bash tools/profile_ta.sh c3      -> gpurun_out/prof/<workload>_ta{1..}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
w=${1:-c3}
k=1
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  rm -rf gpurun_out/prof/${w}_ta$k
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof/${w}_ta$k -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --legs resident > gpurun_out/prof/${w}_ta$k.log 2>&1 || echo "pass $k failed: $set"
  python3 tools/pmc_table.py gpurun_out/prof/${w}_ta$k 2>/dev/null | grep "k_probe"
  k=$((k+1))
done
