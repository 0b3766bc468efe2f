#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/bf16; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o t -- python3 tools/one_config.py 17 6 256 10 2 bf16 1500 > $O/kt.log 2>&1
python tools/rocpd_summary.py $O/kt/t_results.db | head -8
tail -1 $O/kt.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt32 -o t -- python3 tools/one_config.py 17 6 256 10 2 fp32 1500 > $O/kt32.log 2>&1
python tools/rocpd_summary.py $O/kt32/t_results.db | head -6
tail -1 $O/kt32.log
