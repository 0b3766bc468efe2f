cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
true
for v in 0 1 0 1; do
  for cfg in "17 6 512 10 2" "3 1 256 2 2" "17 6 512 2 2" "376 17 512 2 2"; do
    echo -n "slot_by_value=$v : "; SSAC_SLOT_BY_VALUE=$v timeout 300 python tools/one_config.py $cfg fp32 2000 2>&1 | tail -1
  done
done 2>&1 | tee gpurun_out/r4b/slot_rows.log
