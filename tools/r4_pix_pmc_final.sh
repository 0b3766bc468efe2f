#!/bin/bash
# fresh PMC passes of both pixel configurations on the final code (databases deleted after the summaries: 64 MiB limit)
export R4_OUT=gpurun_out/r4/final8
for c in B2 C1; do
  bash tools/prof_r4.sh $c > /dev/null 2>&1
  find $R4_OUT -name "*.db" -delete
  find $R4_OUT -type d -empty -delete
done
ls $R4_OUT; tail -12 $R4_OUT/pmc_dmc_b.md; tail -12 $R4_OUT/pmc_atari_b.md
