#!/bin/bash
O=gpurun_out/r5/bwd_ab; mkdir -p $O
for r in 1 2 3 4; do
  for tag in "" direct staged; do
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] M N10: /" >> $O/rows.txt
  done
done
for tag in "" direct staged; do
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 376 17 512 16 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] Humanoid N16: /" >> $O/rows.txt
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 16 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] M N16: /" >> $O/rows.txt
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 2 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] M 2-of-16: /" >> $O/rows.txt
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 256 10 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] M B256: /" >> $O/rows.txt
done
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_cases.py -x -q -k "chain or redq_M or graph_replay or bench" > $O/pytest.log 2>&1; echo "tests rc $? $(tail -1 $O/pytest.log)" >> $O/rows.txt
cat $O/rows.txt
