#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + the three PMC passes for each workload.
#   bash tools/profile_all.sh c3 c2 c4 c5
# Output under gpurun_out/prof/<workload>_{trace,fetch,write,l2}; summarise afterwards with tools/prof_summary.py.
# Every run uses bench.py's HBM-resident leg only (--legs resident): the kernel averages are those of the roofline leg.
# Counter passes are separate runs (no --stats / tracing domains next to --pmc), the program itself follows `--`.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for w in "$@"; do
  sha256sum mmannot_amd/csrc/mm_device.hip > gpurun_out/prof/${w}_mm_device_sha.txt
  rm -rf gpurun_out/prof/${w}_trace gpurun_out/prof/${w}_fetch gpurun_out/prof/${w}_write gpurun_out/prof/${w}_l2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${w}_trace -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --legs resident > gpurun_out/prof/${w}_trace.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/${w}_fetch -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --legs resident > gpurun_out/prof/${w}_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/${w}_write -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --legs resident > gpurun_out/prof/${w}_write.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof/${w}_l2 -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --legs resident > gpurun_out/prof/${w}_l2.log 2>&1
  grep '^{"metric"' gpurun_out/prof/${w}_trace.log | tail -n 1 > gpurun_out/prof/${w}_bench_under_rocprof.json
  echo "profiled $w"
done
