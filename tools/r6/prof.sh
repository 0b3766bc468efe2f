# round-6 evidence runs (GPU box):
#   gpurun -- 'bash tools/r6/prof.sh A'   headline: kernel trace + separate PMC passes of the bench command, traffic record,
#                                         bench lines (driver form and 2000-step form), config rows, per-rank sharding budget,
#                                         one-device --gpus N --critics 16
#   gpurun -- 'bash tools/r6/prof.sh C'   phase stamps / per-workgroup timelines on the LAB build (./build.sh --lab beforehand)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=${R6_OUT:-gpurun_out/r6/final}
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-secondary"
case "$1" in
A)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o t -- $B --steps 400 --warmup 100 --repeats 3 > $O/kt.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o f -- $B --steps 60 --warmup 20 --repeats 1 > $O/pf.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o w -- $B --steps 60 --warmup 20 --repeats 1 > $O/pw.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pm -o m -- $B --steps 60 --warmup 20 --repeats 1 > $O/pm.log 2>&1
  python tools/rocpd_summary.py $O/kt/t_results.db | head -10 > $O/kernel_trace.md; cat $O/kernel_trace.md
  for p in pf/f pw/w pm/m; do python tools/pmc_summary.py $O/${p}_results.db > $O/pmc_$(basename $p).md 2>&1; tail -6 $O/pmc_$(basename $p).md; done
  python tools/traffic_record.py $O/pf/f_results.db $O/pw/w_results.db "round 6, $(date -u +%Y-%m-%d)" "profiles/r6_kernel_stats.md, profiles/r6_raw/final/pmc_f.md + pmc_w.md" > $O/traffic.log 2>&1; cat $O/traffic.log; cp profiles/traffic.json $O/traffic.json
  rm -rf $O/kt/*.db $O/pf/*.db $O/pw/*.db $O/pm/*.db 2>/dev/null
  timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/b20.err; head -c 900 $O/bench_steps20.json; echo
  timeout 300 python bench.py --steps 2000 --warmup 200 --repeats 5 --no-cpu-baseline --no-secondary > $O/bench_steps2000.json 2> $O/b2000.err; head -c 400 $O/bench_steps2000.json; echo
  timeout 600 python tools/bench_configs.py > $O/configs.md 2>/dev/null; cat $O/configs.md
  timeout 900 python tools/shard_budget.py > $O/shard_budget.md 2>$O/shard_budget.err; cat $O/shard_budget.md
  for n in 2 4 8; do timeout 400 python bench.py --gpus $n --critics 16 --steps 200 --warmup 50 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_gpus${n}_n16_one_device.json 2> $O/bg$n.err; tail -c 700 $O/bench_gpus${n}_n16_one_device.json; echo; done
  ;;
C)
  timeout 300 python tools/fp32_phases.py > $O/phases_M.txt 2>&1; tail -8 $O/phases_M.txt
  timeout 300 python tools/wg_timeline.py 512 10 > $O/wg_timeline_17_512_10.txt 2>&1; grep "^\[2\]\|^      " $O/wg_timeline_17_512_10.txt | tail -12
  ;;
esac
