#!/bin/bash
O=gpurun_out/r5/t2; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "chain_launch or routing" > $O/pytest_kernels.log 2>&1
echo "kernels rc $? $(tail -1 $O/pytest_kernels.log)" >> $O/summary.txt
timeout 900 python -m pytest tests/test_hip_cases.py -x -q -k "recorded_actor or actor_update or engine_matches or graph_replay" > $O/pytest_cases.log 2>&1
echo "cases rc $? $(tail -1 $O/pytest_cases.log)" >> $O/summary.txt
timeout 1200 python -m pytest tests/test_hip_sharded.py -x -q > $O/pytest_sharded.log 2>&1
echo "sharded rc $? $(tail -1 $O/pytest_sharded.log)" >> $O/summary.txt
timeout 600 python -m pytest tests/test_hip_checkpoint.py tests/test_hip_bench_bridge.py -x -q > $O/pytest_misc.log 2>&1
echo "misc rc $? $(tail -1 $O/pytest_misc.log)" >> $O/summary.txt
for r in 1 2 3; do
  timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 3000 2>&1 | tail -1 | sed "s/^/product M N10: /" >> $O/rows.txt
done
timeout 300 python tools/fp32_phases.py 512 10 > $O/phases.txt 2>&1
timeout 900 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench2000.json 2>$O/bench2000.err
cat $O/summary.txt $O/rows.txt; grep "wgrad tile 0" $O/phases.txt | cut -c1-250; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5/t2/bench2000.json"))
print(d["value"], d["roofline"]["frac"])
print(json.dumps(d["secondary"].get("fp32_sweep"))[:1200]); print(d["secondary"].get("error"))
PY
