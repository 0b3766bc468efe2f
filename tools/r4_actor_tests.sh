cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_hip_cases.py -x -q -m gpu -k "actor_update_chained or pendulum_sac or redq_small or sunrise" > gpurun_out/r4b/actor_tests_${1:-a}.log 2>&1; tail -25 gpurun_out/r4b/actor_tests_${1:-a}.log
for v in 0 1; do for cfg in "17 6 512 10" "17 6 256 10" "3 1 256 2"; do echo -n "ACTOR_CHAIN=$v "; SSAC_ACTOR_CHAIN=$v timeout 300 python tools/actor_update_rows.py $cfg 2>&1 | tail -1; done; done | tee gpurun_out/r4b/actor_rows_${1:-a}.log
