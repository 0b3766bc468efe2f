#!/bin/bash
O=gpurun_out/r5/burst; mkdir -p $O
for r in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['repeat_ms_per_step'])" >> $O/rows.txt
  HSA_ENABLE_INTERRUPT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HSA_ENABLE_INTERRUPT=0', d['value'], d['repeat_ms_per_step'])" >> $O/rows.txt
done
timeout 300 python bench.py --steps 2000 --warmup 200 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default 2000', d['value'])" >> $O/rows.txt
HSA_ENABLE_INTERRUPT=0 timeout 300 python bench.py --steps 2000 --warmup 200 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HSA_ENABLE_INTERRUPT=0 2000', d['value'])" >> $O/rows.txt
timeout 900 python tools/shard_budget.py > $O/shard_budget.md 2>&1
cat $O/rows.txt; cat $O/shard_budget.md | tail -12
