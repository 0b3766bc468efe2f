#!/bin/bash
# same-box A/B of two SSAC_DEBUG settings on the current build:  tools/ab/run_env_ab.sh "<A setting>" "<B setting>"
for r in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export SSAC_DEBUG="$1"; else export SSAC_DEBUG="$2"; fi
    python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'steps2000', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
