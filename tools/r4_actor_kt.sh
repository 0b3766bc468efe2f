cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r4b/kt_actor_$1 -o t -- python3 tools/actor_update_rows.py 17 6 512 10 400 > gpurun_out/r4b/kt_actor_$1.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r4b/kt_actor_$1 -name "*.db" | head -1) | head -12 > gpurun_out/r4b/kernel_trace_actor_$1.md
cat gpurun_out/r4b/kernel_trace_actor_$1.md; tail -1 gpurun_out/r4b/kt_actor_$1.log | cut -c1-200
rm -rf gpurun_out/r4b/kt_actor_$1
