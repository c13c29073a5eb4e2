cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q -m gpu > gpurun_out/pytest_r3_init.log 2>&1; tail -3 gpurun_out/pytest_r3_init.log
grep -q " passed" gpurun_out/pytest_r3_init.log && ! grep -q failed gpurun_out/pytest_r3_init.log || exit 1
AB_LIBS="tools/lib_vF.so mmannot_amd/libmmannot_amd.so" bash tools/ab_libs.sh c3 c5 > gpurun_out/ab_init.txt 2>&1; cat gpurun_out/ab_init.txt
