cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
tag=$(echo "$*" | tr ' ' '_')
rocprofv3 --kernel-trace --stats -d gpurun_out/r4b/kt_$tag -o t -- python3 tools/one_config.py $* 800 > gpurun_out/r4b/kt_$tag.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r4b/kt_$tag -name "*.db" | head -1) | head -8 > gpurun_out/r4b/kernel_trace_$tag.md
cat gpurun_out/r4b/kernel_trace_$tag.md; grep "us per critic" gpurun_out/r4b/kt_$tag.log
rm -rf gpurun_out/r4b/kt_$tag
