#!/bin/bash
O=gpurun_out/r5/bfu1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_bf16.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" > $O/summary.txt; tail -3 $O/pytest.log >> $O/summary.txt
for r in 1 2 3; do
  timeout 300 python tools/one_config.py 17 6 256 10 2 bf16 3000 2>&1 | tail -1 >> $O/rows.txt
done
timeout 300 python tools/one_config.py 3 1 256 2 2 bf16 3000 2>&1 | tail -1 >> $O/rows.txt
timeout 300 python tools/one_config.py 17 6 512 10 2 bf16 3000 2>&1 | tail -1 >> $O/rows.txt
cat $O/summary.txt $O/rows.txt
