#!/bin/bash
O=gpurun_out/r5/gpu_suite; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "gpu suite rc $? $(tail -1 $O/pytest_gpu.log)" >> $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $? $(tail -1 $O/smoke.log)" >> $O/summary.txt
cat $O/summary.txt
