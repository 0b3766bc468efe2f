# A/B of two builds of the library inside ONE gpurun call (box-to-box variance is ~1 us per update):
#   tools/ab/A.so, tools/ab/B.so  ->  alternating runs of the headline configuration
# usage on the GPU box: bash tools/ab.sh [rounds]
cd $GRAFT_REPO_ROOT
cp super_sac_amd/libssac_hip.so /tmp/orig.so
for r in $(seq 1 ${1:-3}); do
  for v in A B; do
    cp tools/ab/$v.so super_sac_amd/libssac_hip.so
    echo -n "$v: "; timeout 120 python tools/one_config.py 17 6 512 10 2 ${2:-fp32} 3000 2>/dev/null | tail -1
  done
done
cp /tmp/orig.so super_sac_amd/libssac_hip.so
