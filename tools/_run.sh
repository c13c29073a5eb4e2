cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 tools/soak_probe.py 4000 2000 > gpurun_out/soak_probe.txt 2>&1; grep -c MISMATCH gpurun_out/soak_probe.txt; tail -3 gpurun_out/soak_probe.txt
