#!/bin/bash
# where the actor / temperature updates' time goes (headline shape): host vs wall per call, kernel trace of the actor update
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/actor1; mkdir -p $O
timeout 300 python tools/env_step_phases.py fp32 > $O/env_step_phases.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/akt -o t -- python3 tools/actor_update_rows.py 17 6 512 10 400 > $O/actor_rows.log 2>&1
python tools/rocpd_summary.py /tmp/akt/t_results.db > $O/actor_trace.md 2>&1; rm -rf /tmp/akt
tail -5 $O/env_step_phases.txt; tail -1 $O/actor_rows.log; head -14 $O/actor_trace.md | cut -c1-160
