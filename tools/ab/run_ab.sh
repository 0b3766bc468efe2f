#!/bin/bash
# same-box A/B of two builds of the library (tools/ab/A.so, tools/ab/B.so): bench lines alternated
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for v in A B; do
    cp tools/ab/$v.so super_sac_amd/libssac_hip.so
    python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'steps2000', d['value'], d['ms_per_step'])"
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'steps20  ', d['value'], d['ms_per_step'])"
  done
done
