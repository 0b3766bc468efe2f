#!/bin/bash
O=gpurun_out/r5/t3; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_cases.py -x -q -k "full_size" > $O/pytest_full.log 2>&1
echo "full-size rc $? $(tail -1 $O/pytest_full.log)" >> $O/summary.txt
grep "worst deviations\|Error\|assert" $O/pytest_full.log | head -10 >> $O/summary.txt
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "chain_launch" > $O/pytest_kernels.log 2>&1
echo "kernels rc $? $(tail -1 $O/pytest_kernels.log)" >> $O/summary.txt
timeout 900 python -m pytest tests/test_hip_bf16.py -x -q > $O/pytest_bf16.log 2>&1
echo "bf16 rc $? $(tail -1 $O/pytest_bf16.log)" >> $O/summary.txt
cat $O/summary.txt
