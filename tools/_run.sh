set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c3 c2 c4 c5; do
  timeout -k 10 500 python3 bench.py --workload $w > gpurun_out/r3_bench_${w}_default.json 2> gpurun_out/r3_bench_${w}_default.err || exit 1
done
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_c3_driver_shape.json 2> gpurun_out/r3_bench_c3_driver_shape.err
