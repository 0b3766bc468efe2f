#!/bin/bash
# register-chained bf16 forward: counters of the kernel (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/bf4; mkdir -p $O
export SSAC_BF16_FWD_FORM=1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pm -o m -- python3 tools/bf16_fwd_rows.py > $O/pm.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/pa -o a -- python3 tools/bf16_fwd_rows.py > $O/pa.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $O/pb -o b -- python3 tools/bf16_fwd_rows.py > $O/pb.log 2>&1
for p in pm/m pa/a pb/b; do python tools/pmc_summary.py $O/${p}_results.db bf_regchain > $O/pmc_$(basename $p).md 2>&1; cat $O/pmc_$(basename $p).md; done
