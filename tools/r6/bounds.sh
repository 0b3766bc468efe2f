# Round 6, item 1 of the round-5 review: UPPER BOUNDS of three levers on the headline's two launches, measured on LAB experiment
# builds (wrong values, right timing) before anything is built.  Build container beforehand:
#   ./build.sh --lab; ./build.sh --lab --tag dq -DSSAC_EXP_DQ_READY; ./build.sh --lab --tag fadam -DSSAC_EXP_FAST_ADAM;
#   ./build.sh --lab --tag idx -DSSAC_EXP_IDENTITY_IDX; ./build.sh --lab --tag all3 -DSSAC_EXP_DQ_READY -DSSAC_EXP_FAST_ADAM -DSSAC_EXP_IDENTITY_IDX
# GPU box:  gpurun -- 'bash tools/r6/bounds.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6/bounds; mkdir -p $O
run() {  # tag
  for rep in 1 2; do
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$1 timeout 200 python bench.py --steps 2000 --warmup 300 --repeats 5 --no-cpu-baseline --no-secondary 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.ljust(8) or 'base', 'us/update', round(1e3*d['ms_per_step'],2), 'min', round(1e3*d['ms_per_step_min'],2), 'chained launch us', d['roofline']['avg_launch_us'])"
  done
}
{ SSAC_LAB_TAG= ; for t in "" dq fadam idx all3 "" ; do run "$t"; done; } | tee $O/rows.txt
