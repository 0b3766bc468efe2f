# PMC passes of the pixel configurations (separate passes, kernel trace only):  gpurun -- 'bash tools/prof_pixels_pmc.sh dmc'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C=${1:-dmc}
O=gpurun_out/r2/pmc_$C
mkdir -p $O
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/a -o a -- python3 tools/bench_pixels.py $C 4 > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $O/b -o b -- python3 tools/bench_pixels.py $C 4 > $O/b.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d $O/c -o c -- python3 tools/bench_pixels.py $C 4 > $O/c.log 2>&1
for p in a b c; do python tools/pmc_summary.py $O/$p/${p}_results.db conv; done
