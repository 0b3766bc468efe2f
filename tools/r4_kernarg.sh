cd $GRAFT_REPO_ROOT
for v in 0 1; do
  for cfg in "17 6 512 10 2" "3 1 256 2 2"; do
    echo -n "HIP_FORCE_DEV_KERNARG=$v : "; HIP_FORCE_DEV_KERNARG=$v timeout 300 python tools/one_config.py $cfg fp32 2000 2>&1 | tail -1
  done
done
echo -n "unset: "; timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 2000 2>&1 | tail -1
