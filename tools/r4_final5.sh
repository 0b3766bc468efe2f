#!/bin/bash
# end-of-round evidence: full GPU suite, pixel traces, driver-style + default bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
O=gpurun_out/r4f
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite exit $?" >> $O/gpu_suite.log
tail -4 $O/gpu_suite.log
export TMPDIR=/tmp
for c in dmc atari; do
rocprofv3 --kernel-trace --stats -d $O/kt_pix_$c -o t -- python3 tools/bench_pixels.py $c 20 > $O/kt_pix_$c.log 2>&1
python tools/rocpd_summary.py $(find $O/kt_pix_$c -name "*.db" | head -1) > $O/pix_${c}_trace_final.md
rm -rf $O/kt_pix_$c
tail -1 $O/kt_pix_$c.log | cut -c1-120
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'P'
import json
for f in ("bench_driver", "bench_default"):
    t = open(f"gpurun_out/r4f/{f}.json").read()
    j = json.loads(t[t.index('{"metric"'):])
    s = j["secondary"]
    print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], s["config3_dmc_pixels"]["ms_per_critic_update"], s["config4_atari_pixels"]["ms_per_critic_update"], s["full_redq_step_fp32"]["ms_per_env_step"])
P
