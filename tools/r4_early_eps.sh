cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 1200 python -m pytest tests/test_hip_bench_bridge.py tests/test_hip_bf16.py tests/test_hip_kernels.py tests/test_hip_cases.py -x -q -m gpu -k "bridge or bf16 or chain or sample or redq_small or pendulum or graph_replay or actor" > gpurun_out/r4b/early_eps_tests.log 2>&1; tail -3 gpurun_out/r4b/early_eps_tests.log
for cfg in "3 1 256 2 2 fp32" "17 6 512 2 2 fp32" "17 6 512 4 2 fp32" "376 17 512 2 2 fp32" "376 17 512 4 2 fp32" "17 6 512 10 2 fp32" "17 6 256 10 2 bf16" "3 1 256 2 2 bf16"; do timeout 300 python tools/one_config.py $cfg 2000 2>&1 | tail -1; done | tee gpurun_out/r4b/early_eps_rows.log
for i in 1 2; do timeout 600 python -m pytest tests/test_hip_sharded.py -x -q -m gpu > gpurun_out/r4b/sharded_rep$i.log 2>&1; tail -1 gpurun_out/r4b/sharded_rep$i.log; done
