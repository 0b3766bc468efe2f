#!/bin/bash
# round 5, first co-resident chained launch: parity subset, then A/B rows (form 0 = one workgroup per CU, 1 = co-resident)
mkdir -p gpurun_out/r5/co1
O=gpurun_out/r5/co1
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "chain_launch or fused" > $O/pytest_kernels.log 2>&1
echo "kernels rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_hip_cases.py tests/test_hip_bench_bridge.py -x -q > $O/pytest_cases.log 2>&1
echo "cases rc $?" >> $O/summary.txt
for r in 1 2 3; do
  for f in 0 1; do
    SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/form $f: /" >> $O/rows.txt
  done
done
for f in 0 1; do
  SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 16 2 fp32 2000 2>&1 | tail -1 | sed "s/^/N16 form $f: /" >> $O/rows.txt
  SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 256 10 2 fp32 2000 2>&1 | tail -1 | sed "s/^/B256 form $f: /" >> $O/rows.txt
  SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 6 2 fp32 2000 2>&1 | tail -1 | sed "s/^/N6 form $f: /" >> $O/rows.txt
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench20.json 2>$O/bench20.err
cat $O/summary.txt $O/rows.txt; tail -c 1500 $O/bench20.json
