# round 4: the latency form of the weight-gradient launch -- parity on every fixture, then the under-filled rows
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests/test_hip_cases.py -x -q -k "test_engine_matches_reference" > gpurun_out/r4/small_parity.log 2>&1
echo "parity exit $?" >> gpurun_out/r4/small_parity.log
tail -5 gpurun_out/r4/small_parity.log
rm -f gpurun_out/r4/small_rows.log
for v in 1 0; do
  echo "== SSAC wgrad variant $v" >> gpurun_out/r4/small_rows.log
  for cfg in "3 1 256 2 2" "17 6 512 2 2" "17 6 512 4 2" "17 6 512 8 2" "17 6 512 10 2" "376 17 512 2 2" "376 17 512 4 2"; do
    SSAC_WGRAD_VARIANT=$v timeout 300 python tools/one_config.py $cfg fp32 1500 2>&1 | tail -1 >> gpurun_out/r4/small_rows.log
  done
done
echo "== variant 2 forced at N 8, N 10" >> gpurun_out/r4/small_rows.log
SSAC_WGRAD_VARIANT=2 timeout 300 python tools/one_config.py 17 6 512 8 2 fp32 1500 2>&1 | tail -1 >> gpurun_out/r4/small_rows.log
SSAC_WGRAD_VARIANT=2 timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 1500 2>&1 | tail -1 >> gpurun_out/r4/small_rows.log
cat gpurun_out/r4/small_rows.log
timeout 300 python tools/wg_timeline_small.py 17 6 512 2 > gpurun_out/r4/tl_small_M2.txt 2>&1; tail -12 gpurun_out/r4/tl_small_M2.txt
timeout 300 python tools/wg_timeline_small.py 3 1 256 2 > gpurun_out/r4/tl_small_sac.txt 2>&1; tail -12 gpurun_out/r4/tl_small_sac.txt
