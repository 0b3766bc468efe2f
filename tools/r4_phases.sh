cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 300 python tools/small_phases.py 17 6 512 2 > gpurun_out/r4/small_phases_M2.txt 2>&1; tail -5 gpurun_out/r4/small_phases_M2.txt
timeout 300 python tools/small_phases.py 17 6 512 10 > gpurun_out/r4/small_phases_M10.txt 2>&1; tail -5 gpurun_out/r4/small_phases_M10.txt
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "priority" > gpurun_out/r4/per_tests.log 2>&1; tail -5 gpurun_out/r4/per_tests.log
SSAC_WGRAD_VARIANT=2 timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 1500 2>&1 | tail -1
