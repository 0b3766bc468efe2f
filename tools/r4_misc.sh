cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_hip_sharded.py -x -q > gpurun_out/r4/sharded_tests.log 2>&1; echo "sharded exit $?" >> gpurun_out/r4/sharded_tests.log; tail -4 gpurun_out/r4/sharded_tests.log
for v in 1 0; do echo "== variant $v"; SSAC_WGRAD_VARIANT=$v timeout 300 python tools/env_step_phases.py fp32 2>&1 | tail -4; done > gpurun_out/r4/env_step.txt; cat gpurun_out/r4/env_step.txt
