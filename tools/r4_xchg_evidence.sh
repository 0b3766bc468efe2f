# Failing-first evidence for tests/test_hip_sharded.py::test_owners_only_exchange_survives_a_stalled_non_owner:
# the same test against round 3's library (tools/ab/libssac_r3.so, git-ignored: `git checkout 21bde52 && ./build.sh`),
# then against today's.  Run on the GPU box: bash tools/r4_xchg_evidence.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
cp super_sac_amd/libssac_hip.so /tmp/new.so; cp super_sac_amd/_lib.py /tmp/_lib_new.py
cp tools/ab/libssac_r3.so super_sac_amd/libssac_hip.so
sed -i '/"ssac_xchg_test_mode"/d' super_sac_amd/_lib.py
timeout 600 python -m pytest tests/test_hip_sharded.py -k stalled -x -q > gpurun_out/r4/lap_r3lib.log 2>&1
echo "r3 library: exit $?" >> gpurun_out/r4/lap_r3lib.log
cp /tmp/new.so super_sac_amd/libssac_hip.so; cp /tmp/_lib_new.py super_sac_amd/_lib.py
timeout 900 python -m pytest tests/test_hip_sharded.py -x -q > gpurun_out/r4/lap_newlib.log 2>&1
echo "today's library: exit $?" >> gpurun_out/r4/lap_newlib.log
