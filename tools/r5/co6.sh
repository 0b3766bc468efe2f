#!/bin/bash
O=gpurun_out/r5/co6; mkdir -p $O
for r in 1 2 3; do
  SSAC_CHAIN_FORM=1 timeout 200 python tools/r5/co_stress.py 10 20000 > $O/stress_$r.txt 2>&1; echo "stress run $r rc $? $(tail -1 $O/stress_$r.txt | cut -c1-100)" >> $O/summary.txt
done
SSAC_CHAIN_FORM=0 timeout 200 python tools/r5/co_stress.py 10 20000 > $O/stress_f0.txt 2>&1; echo "stress form 0 rc $? $(tail -1 $O/stress_f0.txt | cut -c1-100)" >> $O/summary.txt
cat $O/summary.txt; grep -h differs $O/*.txt | head
