#!/bin/bash
# round 5: the exchange of a sharded rank as a tail workgroup of the chained launch: parity (sharded suites), then the rank budget
O=gpurun_out/r5/fx1; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_sharded.py -x -q > $O/pytest_sharded.log 2>&1
echo "sharded rc $? $(tail -1 $O/pytest_sharded.log)" >> $O/summary.txt
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "chain_launch" > $O/pytest_kernels.log 2>&1
echo "kernels rc $? $(tail -1 $O/pytest_kernels.log)" >> $O/summary.txt
timeout 900 python tools/shard_budget.py > $O/shard_budget.md 2>$O/shard_budget.err
cat $O/summary.txt; tail -9 $O/shard_budget.md
