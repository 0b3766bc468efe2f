#!/bin/bash
# under-filled shapes: where does a 2-critic update's time go (timelines + phase stamps, lab build)
O=gpurun_out/r5/uf1; mkdir -p $O
for cfg in "512 2" "256 2"; do
  set -- $cfg
  SSAC_CHAIN_FORM=0 timeout 300 python tools/r5/co_timeline.py $1 $2 > $O/timeline_B$1_N$2.txt 2>&1
  timeout 300 python tools/fp32_phases.py $1 $2 > $O/phases_B$1_N$2.txt 2>&1
  timeout 300 python tools/one_config.py 17 6 $1 $2 2 fp32 3000 2>&1 | tail -1 >> $O/rows.txt
done
cat $O/rows.txt; for f in $O/timeline_*.txt $O/phases_*.txt; do echo "== $f"; tail -9 $f | cut -c1-330; done
