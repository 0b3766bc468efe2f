#!/bin/bash
# round 5: backward-data K loop without LDS staging (DirectNN) in every fused tile: parity, then A/B against the staged loop
O=gpurun_out/r5/nn1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "fused or chain or critic or actor" > $O/pytest_kernels.log 2>&1
echo "kernels rc $? $(tail -1 $O/pytest_kernels.log)" >> $O/summary.txt
timeout 900 python -m pytest tests/test_hip_cases.py -x -q > $O/pytest_cases.log 2>&1
echo "cases rc $? $(tail -1 $O/pytest_cases.log)" >> $O/summary.txt
for r in 1 2 3; do
  for tag in "" stagedbwd; do
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/lab[$tag] M N10: /" >> $O/rows.txt
  done
done
for tag in "" stagedbwd; do
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 16 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] M N16: /" >> $O/rows.txt
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 256 2 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] SAC-like N2 B256: /" >> $O/rows.txt
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 376 17 512 16 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] Humanoid N16: /" >> $O/rows.txt
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag timeout 300 python tools/one_config.py 17 6 512 2 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] M 2-of-16: /" >> $O/rows.txt
done
timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/product M N10: /" >> $O/rows.txt
timeout 300 python tools/fp32_phases.py 512 10 > $O/phases.txt 2>&1
cat $O/summary.txt $O/rows.txt; grep "critic WG" $O/phases.txt | cut -c1-300
