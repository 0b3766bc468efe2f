#!/bin/bash
O=gpurun_out/r5/bwd_ab2; mkdir -p $O
for r in 1 2 3; do
  for tag in "" staged; do
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] M N10: /" >> $O/rows.txt
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 376 17 512 16 2 fp32 1500 2>&1 | tail -1 | sed "s/^/lab[$tag] Humanoid N16: /" >> $O/rows.txt
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 2 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] M 2-of-16: /" >> $O/rows.txt
  done
done
for r in 1 2; do timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product steps2000', d['value'], d['roofline']['frac'])" >> $O/rows.txt; done
sort $O/rows.txt | cut -c1-120
