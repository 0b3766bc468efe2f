cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r4/gpu_suite2.log 2>&1; echo "suite exit $?" >> gpurun_out/r4/gpu_suite2.log; tail -26 gpurun_out/r4/gpu_suite2.log
