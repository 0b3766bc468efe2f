#!/bin/bash
# round 5: co-resident form with direct weight fragments, after the base-pointer fix: stress, parity, rows, timelines
O=gpurun_out/r5/co7; mkdir -p $O
for r in 1 2 3; do
  SSAC_CHAIN_FORM=1 timeout 200 python tools/r5/co_stress.py 10 5000 > $O/stress_$r.txt 2>&1; echo "stress run $r rc $? $(tail -1 $O/stress_$r.txt | cut -c1-100)" >> $O/summary.txt
done
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "chain_launch" > $O/pytest_kernels.log 2>&1
echo "kernels rc $?" >> $O/summary.txt
for r in 1 2 3; do
  for f in 0 1; do
    SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/form $f: /" >> $O/rows.txt
  done
done
for f in 0 1; do
  SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 6 2 fp32 2000 2>&1 | tail -1 | sed "s/^/N6 form $f: /" >> $O/rows.txt
  SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 256 16 2 fp32 2000 2>&1 | tail -1 | sed "s/^/B256 N16 form $f: /" >> $O/rows.txt
  SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 8 2 fp32 2000 2>&1 | tail -1 | sed "s/^/N8 form $f: /" >> $O/rows.txt
done
SSAC_CHAIN_FORM=1 timeout 300 python tools/r5/co_timeline.py 512 10 > $O/timeline_form1.txt 2>&1
SSAC_CHAIN_FORM=1 timeout 300 python tools/fp32_phases.py 512 10 > $O/phases_form1.txt 2>&1
cat $O/summary.txt $O/rows.txt; tail -9 $O/timeline_form1.txt; tail -9 $O/phases_form1.txt | cut -c1-300
