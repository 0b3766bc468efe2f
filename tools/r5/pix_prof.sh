#!/bin/bash
# round 5: kernel traces + PMC passes of the two pixel configurations with the round's final library; only the summaries are kept
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/pix; mkdir -p $O
W=/tmp/pixprof; mkdir -p $W
for c in dmc atari; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $W/pix_$c -o t -- python3 tools/bench_pixels.py $c 20 > $O/pix_$c.log 2>&1
  python tools/rocpd_summary.py $W/pix_$c/t_results.db > $O/pix_${c}_trace.md 2>&1
  P=$W/pmc_$c; mkdir -p $P
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $P/a -o a -- python3 tools/bench_pixels.py $c 4 > $O/pmc_${c}_a.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $P/b -o b -- python3 tools/bench_pixels.py $c 4 > $O/pmc_${c}_b.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/f -o f -- python3 tools/bench_pixels.py $c 4 > $O/pmc_${c}_f.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/w -o w -- python3 tools/bench_pixels.py $c 4 > $O/pmc_${c}_w.log 2>&1
  for p in a b f w; do python tools/pmc_summary.py $P/$p/${p}_results.db conv > $O/pmc_${c}_$p.md 2>&1; done
  head -16 $O/pix_${c}_trace.md | cut -c1-150; tail -1 $O/pix_$c.log
done
rm -rf $W
