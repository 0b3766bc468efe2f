# A/B of two builds (tools/ab/A.so, B.so) on the pixel configurations inside ONE gpurun call
# usage on the GPU box: bash tools/ab_pixels.sh [rounds] [dmc|atari ...]
cd $GRAFT_REPO_ROOT
cp super_sac_amd/libssac_hip.so /tmp/orig.so
R=${1:-2}; shift
timeout 100 python3 tools/bench_pixels.py dmc 5 > /dev/null 2>&1   # first run on a fresh box is slow
for r in $(seq 1 $R); do
  for v in A B; do
    cp tools/ab/$v.so super_sac_amd/libssac_hip.so
    for c in ${@:-dmc atari}; do echo -n "$v: "; timeout 200 python3 tools/bench_pixels.py $c 20 2>/dev/null | tail -1; done
  done
done
cp /tmp/orig.so super_sac_amd/libssac_hip.so
