#!/bin/bash
# round 5: where does the co-resident chained launch spend its time?  lab build: timelines + stamps; experiment variants without
# the MFMAs / without the weight loads / without both (wrong results, timing only)
O=gpurun_out/r5/co2; mkdir -p $O
for f in 0 1; do
  SSAC_CHAIN_FORM=$f timeout 300 python tools/r5/co_timeline.py 512 10 > $O/timeline_form$f.txt 2>&1
  SSAC_CHAIN_FORM=$f timeout 300 python tools/fp32_phases.py 512 10 > $O/phases_form$f.txt 2>&1
done
for tag in "" nomfma nowload neither; do
  for f in 0 1; do
    SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag SSAC_CHAIN_FORM=$f timeout 300 python tools/one_config.py 17 6 512 10 2 fp32 2000 2>&1 | tail -1 | sed "s/^/lab[$tag] form $f: /" >> $O/rows.txt
  done
done
cat $O/rows.txt; cat $O/timeline_form1.txt | tail -12; cat $O/phases_form1.txt | tail -8; cat $O/timeline_form0.txt | tail -8
