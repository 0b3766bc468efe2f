#!/bin/bash
O=gpurun_out/r5/co4; mkdir -p $O
export SSAC_CHAIN_FORM=1
AMD_SERIALIZE_KERNEL=3 timeout 120 python tools/r5/co_debug.py 512 10 0 > $O/eager_serial.txt 2>&1; echo "eager serial rc $?" >> $O/summary.txt
timeout 120 python tools/r5/co_debug.py 512 10 0 > $O/eager.txt 2>&1; echo "eager rc $?" >> $O/summary.txt
timeout 120 python tools/r5/co_debug.py 512 10 0 0 > $O/eager_nofold.txt 2>&1; echo "eager no-fold-gather rc $?" >> $O/summary.txt
timeout 120 python tools/r5/co_debug.py 512 10 1 > $O/graphs.txt 2>&1; echo "recorded rc $?" >> $O/summary.txt
timeout 120 python tools/r5/co_debug.py 256 16 1 > $O/graphs_b256.txt 2>&1; echo "recorded B256 N16 rc $?" >> $O/summary.txt
cat $O/summary.txt; for f in eager_serial eager eager_nofold graphs; do echo "== $f"; tail -4 $O/$f.txt; done
