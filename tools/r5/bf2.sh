#!/bin/bash
# register-chained bf16 forward: what bounds it (experiment builds, wrong results)
O=gpurun_out/r5/bf3; mkdir -p $O
for tag in "" rcnofrag rcnoepi rcnone; do
  SSAC_LAB_BUILD=1 SSAC_LAB_TAG=$tag SSAC_BF16_FWD_FORM=1 timeout 300 python tools/bf16_fwd_rows.py 2>&1 | grep "B65536" | sed "s/^/lab[$tag]: /" >> $O/rows.txt
done
