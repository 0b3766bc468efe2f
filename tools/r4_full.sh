# round 4: the whole GPU suite, then the under-filled rows with today's library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r4/gpu_suite.log 2>&1
echo "suite exit $?" >> gpurun_out/r4/gpu_suite.log
tail -15 gpurun_out/r4/gpu_suite.log
rm -f gpurun_out/r4/small_rows2.log
for cfg in "3 1 256 2 2" "17 6 512 2 2" "17 6 512 4 2" "376 17 512 2 2"; do
  timeout 300 python tools/one_config.py $cfg fp32 1500 2>&1 | tail -1 >> gpurun_out/r4/small_rows2.log
done
cat gpurun_out/r4/small_rows2.log
timeout 300 python tools/wg_timeline_small.py 17 6 512 2 > gpurun_out/r4/tl_small_M2b.txt 2>&1; tail -8 gpurun_out/r4/tl_small_M2b.txt
