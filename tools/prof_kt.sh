# kernel trace of the headline bench command:   bash tools/prof_kt.sh <tag>   (GPU box) -> gpurun_out/<tag>/kt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3}
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o t -- python3 bench.py --steps 400 --warmup 100 --repeats 3 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1
python tools/rocpd_summary.py $O/kt/t_results.db | head -9
tail -c 400 $O/kt.log | grep -o '"value": [0-9.]*'
