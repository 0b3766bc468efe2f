#!/bin/bash
O=gpurun_out/r5/co9; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "chain_launch" > $O/pytest_kernels.log 2>&1
echo "kernels rc $?" >> $O/summary.txt
for r in 1 2; do
  for f in 0 1; do
    SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/form $f: /" >> $O/rows.txt
  done
done
SSAC_CHAIN_FORM=1 timeout 300 python tools/r5/co_timeline.py 512 10 > $O/timeline_form1.txt 2>&1
SSAC_CHAIN_FORM=1 timeout 300 python tools/fp32_phases.py 512 10 > $O/phases_form1.txt 2>&1
cat $O/summary.txt $O/rows.txt; tail -7 $O/timeline_form1.txt; tail -9 $O/phases_form1.txt | head -5 | cut -c1-300
