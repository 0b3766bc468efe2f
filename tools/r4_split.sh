cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_hip_cases.py -x -q -k "test_chained_launch_forms or redq_M or pendulum or test_graph_replay" > gpurun_out/r4/split_parity.log 2>&1; echo "parity exit $?" >> gpurun_out/r4/split_parity.log; tail -6 gpurun_out/r4/split_parity.log
timeout 900 python -m pytest tests/test_hip_sharded.py tests/test_hip_bench_bridge.py -x -q > gpurun_out/r4/split_sharded.log 2>&1; echo "exit $?" >> gpurun_out/r4/split_sharded.log; tail -4 gpurun_out/r4/split_sharded.log
rm -f gpurun_out/r4/split_rows.log
for sp in 0 1; do
  echo "== CHAIN_SPLIT=$sp" >> gpurun_out/r4/split_rows.log
  for cfg in "3 1 256 2 2" "17 6 512 2 2" "17 6 512 4 2" "17 6 256 10 2" "17 6 512 10 2" "376 17 512 2 2" "376 17 512 4 2"; do
    SSAC_CHAIN_SPLIT=$sp timeout 300 python tools/one_config.py $cfg fp32 1500 2>&1 | tail -1 >> gpurun_out/r4/split_rows.log
  done
done
cat gpurun_out/r4/split_rows.log
