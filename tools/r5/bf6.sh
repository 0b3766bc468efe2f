#!/bin/bash
O=gpurun_out/r5/bf6; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_bf16.py -x -q -k "large_batch or ensemble_q_forward" > $O/pytest.log 2>&1; echo "pytest rc $?" > $O/summary.txt
for f in 1 0 1; do
  SSAC_BF16_FWD_FORM=$f timeout 300 python tools/bf16_fwd_rows.py 2>&1 | grep -v amdgpu.ids | sed "s/^/form $f: /" >> $O/rows.txt
done
timeout 300 python tools/r5/rc_phases.py 65536 > $O/phases_65536.txt 2>&1
cat $O/summary.txt; tail -3 $O/pytest.log; cut -c1-125 $O/rows.txt; tail -2 $O/phases_65536.txt
