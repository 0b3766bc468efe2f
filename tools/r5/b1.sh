#!/bin/bash
O=gpurun_out/r5/b1; mkdir -p $O
s=$(date +%s)
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
echo "bench rc $? wall $(( $(date +%s) - s )) s"
python - <<PY
import json
d=json.loads(open("$O/bench20.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
s=d["secondary"]
print(json.dumps(s["config2_bf16"])[:1700])
for k,v in s.items():
    if k!="config2_bf16": print(k, json.dumps(v)[:300])
PY
