# The LAB leg of the GPU suite (GPU box; `./build.sh --lab` in the build container beforehand -- the .so travels):
# the tests whose protocol-replay assertions need entry points that only the lab library defines
# (include/ssac_hip_test.h, "lab hooks": ssac_xchg_test_mode -- round 3's exchange protocol on today's kernel, and a lap
# that must be DETECTED).  Against the product library those assertions are skipped by design.
#   gpurun -- 'bash tools/gpu_suite_lab.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=${LAB_OUT:-gpurun_out/r6/lab}
mkdir -p $O
SSAC_LAB_BUILD=1 timeout 900 python -m pytest tests/test_hip_sharded.py -q -x -k "stalled_non_owner or sharded_sequence or one_shot or gives_up" > $O/lab_suite.log 2>&1
tail -5 $O/lab_suite.log
