cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_hip_bf16.py -x -q -m gpu > gpurun_out/r4b/bf16_tests_$1.log 2>&1; tail -5 gpurun_out/r4b/bf16_tests_$1.log
for v in 1 1; do for cfg in "17 6 256 10 2" "17 6 512 10 2" "3 1 256 2 2" "17 6 512 16 2"; do echo -n "CHAIN_PC=$v "; SSAC_CHAIN_PC=$v timeout 300 python tools/one_config.py $cfg bf16 1500 2>&1 | tail -1; done; done | tee gpurun_out/r4b/bf16_rows_$1.log
