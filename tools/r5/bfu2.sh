#!/bin/bash
O=gpurun_out/r5/bfu5; mkdir -p $O
for r in 1 2 3; do for tag in old ""; do
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 256 10 2 bf16 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] /" >> $O/rows.txt
done; done
for tag in old ""; do
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/bf16_phases.py 256 10 2>&1 | grep "wgrad tile" | sed "s/^/lab[$tag] /" >> $O/rows.txt
done
sort $O/rows.txt | cut -c1-200
