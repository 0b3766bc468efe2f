#!/bin/bash
# the round's final evidence run: tools/prof_r5.sh A with the raw rocprof directories removed afterwards (summaries only)
export R5_OUT=gpurun_out/r5/final2
bash tools/prof_r5.sh A > gpurun_out/r5_final2.log 2>&1
rm -rf $R5_OUT/kt $R5_OUT/pf $R5_OUT/pw $R5_OUT/pm $R5_OUT/kt_sacbf
tail -40 gpurun_out/r5_final2.log | cut -c1-220
