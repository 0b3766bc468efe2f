#!/bin/bash
O=gpurun_out/r5/co5; mkdir -p $O
export SSAC_CHAIN_FORM=1
for tag in "" nonn nofwd noprio nostore; do
  for r in 1 2; do
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 120 python tools/one_config.py 17 6 512 10 2 fp32 3000 > $O/run_$tag$r.txt 2>&1; echo "lab[$tag] run $r rc $? $(tail -1 $O/run_$tag$r.txt | cut -c1-100)" >> $O/summary.txt
  done
done
cat $O/summary.txt
