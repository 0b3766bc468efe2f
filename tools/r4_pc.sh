cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_hip_cases.py -x -q -k "test_chained_launch_forms" > gpurun_out/r4/pc_parity.log 2>&1; echo "parity exit $?" >> gpurun_out/r4/pc_parity.log; tail -12 gpurun_out/r4/pc_parity.log
rm -f gpurun_out/r4/pc_rows.log
for pc in 0 1; do
  echo "== CHAIN_PC=$pc" >> gpurun_out/r4/pc_rows.log
  for cfg in "3 1 256 2 2" "17 6 512 2 2" "17 6 512 4 2" "17 6 512 8 2" "17 6 512 10 2" "17 6 512 16 2" "376 17 512 2 2" "376 17 512 4 2" "376 17 512 16 2"; do
    SSAC_CHAIN_PC=$pc timeout 300 python tools/one_config.py $cfg fp32 1500 2>&1 | tail -1 >> gpurun_out/r4/pc_rows.log
  done
done
cat gpurun_out/r4/pc_rows.log
