cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
./build.sh --lab > /dev/null 2>&1
for cfg in "17 6 512 2" "3 1 256 2" "376 17 512 2" "17 6 512 10"; do
  set -- $cfg
  timeout 300 python tools/wg_timeline_small.py $cfg > gpurun_out/r4/tl_pc_$1_$3_$4.txt 2>&1; grep "^\[2\]\|^      " gpurun_out/r4/tl_pc_$1_$3_$4.txt
done
