cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4b/gpu_suite_$1.log 2>&1; echo "suite exit $?" >> gpurun_out/r4b/gpu_suite_$1.log
tail -5 gpurun_out/r4b/gpu_suite_$1.log
python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | head -20 | tee gpurun_out/r4b/configs_$1.md
