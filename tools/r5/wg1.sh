#!/bin/bash
# weight-gradient launch: upper bound of the K-group off-load (fc2 tiles without their last K iteration; lab build, wrong results)
O=gpurun_out/r5/wg1; mkdir -p $O
for r in 1 2 3; do
  for tag in "" skipiter; do
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] M N10: /" >> $O/rows.txt
  done
done
for tag in "" skipiter; do
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/wg_timeline.py 512 10 > $O/timeline_$tag.txt 2>&1
done
sort $O/rows.txt | cut -c1-120; tail -12 $O/timeline_.txt | cut -c1-250; tail -12 $O/timeline_skipiter.txt | cut -c1-250
