cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/final3
mkdir -p $O
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/b20.err; tail -c 200 $O/bench_steps20.json; echo
timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary > $O/bench_steps2000.json 2> $O/b2000.err; head -c 300 $O/bench_steps2000.json; echo
kt() { # tag, then one_config args
  tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o t -- python3 tools/one_config.py $* 1500 > $O/kt_$tag.log 2>&1
  python tools/rocpd_summary.py $O/kt_$tag/t_results.db | head -6 > $O/kernel_trace_$tag.md; cat $O/kernel_trace_$tag.md; grep "us per critic" $O/kt_$tag.log
  rm -rf $O/kt_$tag
}
kt m2 17 6 512 2 2 fp32
kt sac 3 1 256 2 2 fp32
kt s2 376 17 512 2 2 fp32
kt c2_bf16 17 6 256 10 2 bf16
kt sac_bf16 3 1 256 2 2 bf16
timeout 500 python tools/shard_budget.py > $O/shard_budget.md 2>/dev/null; cat $O/shard_budget.md
for cfg in "17 6 512 10" "17 6 256 10" "3 1 256 2"; do timeout 300 python tools/actor_update_rows.py $cfg 2>&1 | tail -1; done | tee $O/actor_rows.log
