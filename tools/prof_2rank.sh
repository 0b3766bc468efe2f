# kernel trace of the two-rank sharded bench (both ranks on the one GPU of the box): exchange kernel durations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2/kt2
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o t -- python3 bench.py --gpus 2 --steps 300 --warmup 50 --repeats 1 --no-cpu-baseline --no-secondary > gpurun_out/r2/kt2.log 2>&1
ls -R $O | head -20
for f in $(find $O -name "*_results.db"); do echo "== $f"; python tools/rocpd_summary.py $f | head -12; done
tail -c 400 gpurun_out/r2/kt2.log
