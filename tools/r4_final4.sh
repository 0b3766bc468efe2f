mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "first_layer or conv" > $O/conv_test.log 2>&1; echo "tests exit $?" >> $O/conv_test.log
tail -3 $O/conv_test.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'P'
import json
for f in ("bench_driver", "bench_default"):
    t = open(f"gpurun_out/r4c/{f}.json").read()
    j = json.loads(t[t.index('{"metric"'):])
    print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_us"], j.get("cpu_baseline", {}).get("value"))
    s = j.get("secondary", {})
    print({k: (v.get("ms_per_update") if isinstance(v, dict) else v) for k, v in s.items()} if isinstance(s, dict) else s)
P
