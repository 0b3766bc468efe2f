cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
for cfg in "17 6 512 10" "17 6 256 10" "3 1 256 2" "376 17 512 16"; do timeout 300 python tools/actor_update_rows.py $cfg 2>&1 | tail -1; done | tee gpurun_out/r4b/actor_rows_$1.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r4b/kt_actor_$1 -o t -- python3 tools/actor_update_rows.py 17 6 512 10 400 > gpurun_out/r4b/kt_actor_$1.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r4b/kt_actor_$1 -name "*.db" | head -1) | head -10 > gpurun_out/r4b/kernel_trace_actor_$1.md
cat gpurun_out/r4b/kernel_trace_actor_$1.md
rm -rf gpurun_out/r4b/kt_actor_$1
