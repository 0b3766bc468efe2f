#!/bin/bash
O=gpurun_out/r5/full1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "gpu suite rc $? $(tail -1 $O/pytest_gpu.log)" >> $O/summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2>$O/bench20.err
echo "bench rc $?" >> $O/summary.txt
cat $O/summary.txt; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5/full1/bench20.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
s=d.get("secondary",{})
for k in ("scaling_anchors_n16_1gpu","fp32_sweep","full_redq_step_fp32","error"):
    print(k, json.dumps(s.get(k))[:900])
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!="workload" and not isinstance(vv,dict)}) for k,v in s.items() if k.startswith("config")})
print(d.get("cpu_baseline",{}).get("value"))
PY
