cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests/test_hip_cases.py -x -q -k "test_engine_matches_reference and 64x32" > gpurun_out/r4/v3_parity.log 2>&1
echo "parity exit $?" >> gpurun_out/r4/v3_parity.log; tail -3 gpurun_out/r4/v3_parity.log
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "priority" > gpurun_out/r4/per_tests.log 2>&1; tail -3 gpurun_out/r4/per_tests.log
rm -f gpurun_out/r4/v3_rows.log
for v in 1 3 0; do
  echo "== variant $v" >> gpurun_out/r4/v3_rows.log
  for cfg in "17 6 512 4 2" "17 6 512 8 2" "17 6 512 10 2" "17 6 256 10 2" "376 17 512 4 2" "376 17 512 8 2"; do
    SSAC_WGRAD_VARIANT=$v timeout 300 python tools/one_config.py $cfg fp32 1500 2>&1 | tail -1 >> gpurun_out/r4/v3_rows.log
  done
done
cat gpurun_out/r4/v3_rows.log
