# round-3 evidence run (GPU box): kernel trace + PMC passes of the headline bench command, the driver-style bench line,
# the configuration rows, the per-rank sharding budget and the pixel traces.   gpurun -- 'bash tools/prof_r3.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o t -- $B --steps 400 --warmup 100 --repeats 3 > $O/kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o f -- $B --steps 60 --warmup 20 --repeats 1 > $O/pf.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o w -- $B --steps 60 --warmup 20 --repeats 1 > $O/pw.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pm -o m -- $B --steps 60 --warmup 20 --repeats 1 > $O/pm.log 2>&1
python tools/rocpd_summary.py $O/kt/t_results.db | head -10 > $O/kernel_trace.md; cat $O/kernel_trace.md
for p in pf/f pw/w pm/m; do python tools/pmc_summary.py $O/${p}_results.db > $O/pmc_$(basename $p).md 2>&1; tail -6 $O/pmc_$(basename $p).md; done
timeout 400 python bench.py --steps 20 --warmup 5 > $O/b20.json 2> $O/b20.err; tail -c 300 $O/b20.json
timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary > $O/b2000.json 2> $O/b2000.err; head -c 400 $O/b2000.json
timeout 500 python tools/bench_configs.py > $O/configs.md 2>/dev/null; cat $O/configs.md
timeout 400 python tools/shard_budget.py > $O/shard_budget.md 2>/dev/null; cat $O/shard_budget.md
for c in dmc atari; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/pix_$c -o t -- python3 tools/bench_pixels.py $c 20 > $O/pix_$c.log 2>&1
  python tools/rocpd_summary.py $O/pix_$c/t_results.db > $O/pix_${c}_trace.md; head -14 $O/pix_${c}_trace.md; tail -1 $O/pix_$c.log
done
