#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/final_check; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o t -- python3 bench.py --no-cpu-baseline --no-secondary --steps 400 --warmup 100 --repeats 3 > $O/kt.log 2>&1
python tools/rocpd_summary.py $O/kt/t_results.db | head -6 > $O/kernel_trace.md; cat $O/kernel_trace.md
for r in 1 2; do timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps2000', d['value'], d['roofline']['frac'])"; done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null; python -c "import json; d=json.load(open('$O/bench_steps20.json')); print('steps20', d['value'], d['roofline']['frac'], d['secondary'].get('error'))"
