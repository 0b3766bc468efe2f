#!/bin/bash
# round 5: the co-resident chained launch with weight fragments straight from L2 (no LDS staging): parity, A/B rows, timelines
O=gpurun_out/r5/co3; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "chain_launch" > $O/pytest_kernels.log 2>&1
echo "kernels rc $?" >> $O/summary.txt
for r in 1 2; do
  for f in 0 1; do
    SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/form $f: /" >> $O/rows.txt
  done
done
for f in 0 1; do
  SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 6 2 fp32 2000 2>&1 | tail -1 | sed "s/^/N6 form $f: /" >> $O/rows.txt
  SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 256 16 2 fp32 2000 2>&1 | tail -1 | sed "s/^/B256 N16 form $f: /" >> $O/rows.txt
done
for f in 0 1; do
  SSAC_CHAIN_FORM=$f timeout 300 python tools/r5/co_timeline.py 512 10 > $O/timeline_form$f.txt 2>&1
  SSAC_CHAIN_FORM=$f timeout 300 python tools/fp32_phases.py 512 10 > $O/phases_form$f.txt 2>&1
done
for tag in nomfma nowload neither; do
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag SSAC_CHAIN_FORM=1 timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] form 1: /" >> $O/rows.txt
done
cat $O/summary.txt $O/rows.txt; tail -12 $O/timeline_form1.txt; tail -9 $O/phases_form1.txt | cut -c1-400; tail -5 $O/pytest_kernels.log
