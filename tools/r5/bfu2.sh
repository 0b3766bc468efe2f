#!/bin/bash
# A/B of bf16-update variants in one box: LAB build of the working tree against a LAB build tagged `old` (./build.sh --lab --tag old
# from the previous commit), interleaved
O=gpurun_out/r5/bfu6; mkdir -p $O
for r in 1 2 3; do for tag in old ""; do
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 256 10 2 bf16 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] /" >> $O/rows.txt
done; done
for tag in old ""; do
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 3 1 256 2 2 bf16 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] /" >> $O/rows.txt
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 10 2 bf16 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] /" >> $O/rows.txt
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/r5/bf16_timeline.py 256 10 2>&1 | grep -v amdgpu | tail -9 | head -4 | sed "s/^/lab[$tag] /" >> $O/rows.txt
done
sort $O/rows.txt | cut -c1-200
