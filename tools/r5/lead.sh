#!/bin/bash
O=gpurun_out/r5/lead; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "chain_launch" > $O/pytest_kernels.log 2>&1
echo "kernels rc $? $(tail -1 $O/pytest_kernels.log)" >> $O/summary.txt
timeout 600 python -m pytest tests/test_hip_cases.py -x -q -k "redq_M or chained or graph_replay or bench" > $O/pytest_cases.log 2>&1
echo "cases rc $? $(tail -1 $O/pytest_cases.log)" >> $O/summary.txt
for r in 1 2 3 4; do
  for tag in "" prev; do
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/lab[$tag]: /" >> $O/rows.txt
  done
done
for tag in "" prev; do
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 256 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] B256: /" >> $O/rows.txt
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 8 2 fp32 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] N8: /" >> $O/rows.txt
done
timeout 200 python tools/wg_timeline.py 512 10 > $O/timeline.txt 2>&1
cat $O/summary.txt $O/rows.txt; grep "^\[2\]\|^      [pc]" $O/timeline.txt | head -8
