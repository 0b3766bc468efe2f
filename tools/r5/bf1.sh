#!/bin/bash
# bf16 ensemble-Q forward, large batches: register-chained kernel against the streaming kernel
O=gpurun_out/r5/bf1; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_bf16.py -x -q -k "large_batch or ensemble_q_forward" > $O/pytest.log 2>&1; echo "pytest rc $?" > $O/summary.txt
for r in 1 2; do for f in 0 1; do
  SSAC_BF16_FWD_FORM=$f timeout 300 python tools/bf16_fwd_rows.py 2>&1 | grep -v amdgpu.ids | sed "s/^/form $f: /" >> $O/rows.txt
done; done
cat $O/summary.txt; tail -5 $O/pytest.log; cat $O/rows.txt
