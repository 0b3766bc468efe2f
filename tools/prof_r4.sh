# round-4 evidence runs (GPU box), in five calls (a call may bring back at most 64 MiB):
#   gpurun -- 'bash tools/prof_r4.sh A'    headline: kernel trace + PMC passes of the bench command, bench lines, config
#                                          rows, per-rank sharding budget, one-device --gpus N value checks
#   gpurun -- 'bash tools/prof_r4.sh B1'   under-filled kernel traces (2-of-16 at M and Humanoid, SAC fp32 / bf16)
#   gpurun -- 'bash tools/prof_r4.sh B2'   DMC pixel trace + PMC passes
#   gpurun -- 'bash tools/prof_r4.sh C1'   Atari pixel trace + PMC passes
#   gpurun -- 'bash tools/prof_r4.sh C2'   phase stamps / per-workgroup timelines on the LAB build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=${R4_OUT:-gpurun_out/r4/final}
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-secondary"
pix() {
  c=$1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/pix_$c -o t -- python3 tools/bench_pixels.py $c 20 > $O/pix_$c.log 2>&1
  python tools/rocpd_summary.py $O/pix_$c/t_results.db > $O/pix_${c}_trace.md; head -14 $O/pix_${c}_trace.md; tail -1 $O/pix_$c.log
  P=$O/pmc_$c; mkdir -p $P
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $P/a -o a -- python3 tools/bench_pixels.py $c 4 > $P/a.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $P/b -o b -- python3 tools/bench_pixels.py $c 4 > $P/b.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/f -o f -- python3 tools/bench_pixels.py $c 4 > $P/f.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/w -o w -- python3 tools/bench_pixels.py $c 4 > $P/w.log 2>&1
  for p in a b f w; do python tools/pmc_summary.py $P/$p/${p}_results.db conv > $O/pmc_${c}_$p.md 2>&1; tail -8 $O/pmc_${c}_$p.md; done
}
case "$1" in
A)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o t -- $B --steps 400 --warmup 100 --repeats 3 > $O/kt.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o f -- $B --steps 60 --warmup 20 --repeats 1 > $O/pf.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o w -- $B --steps 60 --warmup 20 --repeats 1 > $O/pw.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pm -o m -- $B --steps 60 --warmup 20 --repeats 1 > $O/pm.log 2>&1
  python tools/rocpd_summary.py $O/kt/t_results.db | head -10 > $O/kernel_trace.md; cat $O/kernel_trace.md
  for p in pf/f pw/w pm/m; do python tools/pmc_summary.py $O/${p}_results.db > $O/pmc_$(basename $p).md 2>&1; tail -6 $O/pmc_$(basename $p).md; done
  timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/b20.err; tail -c 300 $O/bench_steps20.json
  timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary > $O/bench_steps2000.json 2> $O/b2000.err; head -c 400 $O/bench_steps2000.json
  timeout 600 python tools/bench_configs.py > $O/configs.md 2>/dev/null; cat $O/configs.md
  timeout 500 python tools/shard_budget.py > $O/shard_budget.md 2>/dev/null; cat $O/shard_budget.md
  for n in 2 4 8; do timeout 400 python bench.py --gpus $n --steps 200 --warmup 50 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_gpus${n}_one_device.json 2> $O/bg$n.err; tail -c 700 $O/bench_gpus${n}_one_device.json; echo; done
  ;;
B1)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_m2 -o t -- python3 tools/one_config.py 17 6 512 2 2 fp32 1500 > $O/kt_m2.log 2>&1
  python tools/rocpd_summary.py $O/kt_m2/t_results.db | head -8 > $O/kernel_trace_m2.md; cat $O/kernel_trace_m2.md
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_sac -o t -- python3 tools/one_config.py 3 1 256 2 2 fp32 1500 > $O/kt_sac.log 2>&1
  python tools/rocpd_summary.py $O/kt_sac/t_results.db | head -8 > $O/kernel_trace_sac.md; cat $O/kernel_trace_sac.md
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_s2 -o t -- python3 tools/one_config.py 376 17 512 2 2 fp32 1500 > $O/kt_s2.log 2>&1
  python tools/rocpd_summary.py $O/kt_s2/t_results.db | head -8 > $O/kernel_trace_s2.md; cat $O/kernel_trace_s2.md
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_sacbf -o t -- python3 tools/one_config.py 3 1 256 2 2 bf16 1500 > $O/kt_sacbf.log 2>&1
  python tools/rocpd_summary.py $O/kt_sacbf/t_results.db | head -8 > $O/kernel_trace_sac_bf16.md; cat $O/kernel_trace_sac_bf16.md; tail -1 $O/kt_sacbf.log
  ;;
B2)
  pix dmc
  ;;
C1)
  pix atari
  ;;
C2)
  ./build.sh --lab > /dev/null 2>&1
  timeout 300 python tools/fp32_phases.py > $O/phases_M.txt 2>&1; tail -8 $O/phases_M.txt
  timeout 300 python tools/small_phases.py 17 6 512 2 > $O/small_phases_M2.txt 2>&1; tail -4 $O/small_phases_M2.txt
  for cfg in "17 6 512 2" "3 1 256 2" "376 17 512 2" "17 6 512 10"; do
    set -- $cfg
    timeout 300 python tools/wg_timeline_small.py $cfg > $O/wg_timeline_$1_$3_$4.txt 2>&1; grep "^\[2\]\|^      " $O/wg_timeline_$1_$3_$4.txt
  done
  ;;
esac
