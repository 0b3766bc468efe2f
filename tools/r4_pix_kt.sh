cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r4b/kt_pix_$1 -o t -- python3 tools/bench_pixels.py $1 20 > gpurun_out/r4b/kt_pix_$1.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r4b/kt_pix_$1 -name "*.db" | head -1) > gpurun_out/r4b/pix_$1_trace_$2.md
head -24 gpurun_out/r4b/pix_$1_trace_$2.md; tail -2 gpurun_out/r4b/kt_pix_$1.log | cut -c1-150
rm -rf gpurun_out/r4b/kt_pix_$1
