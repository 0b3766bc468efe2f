cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 400 python tools/shard_budget.py > gpurun_out/r2/shard_budget.md 2>/dev/null; cat gpurun_out/r2/shard_budget.md
timeout 300 python bench.py --gpus 2 --steps 1000 --warmup 200 --repeats 3 --no-cpu-baseline --no-secondary > gpurun_out/r2/b2.json 2> gpurun_out/r2/b2.err; tail -c 1500 gpurun_out/r2/b2.json
timeout 500 python tools/bench_configs.py > gpurun_out/r2/configs.md 2>/dev/null; cat gpurun_out/r2/configs.md
