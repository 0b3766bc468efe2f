cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -d gpurun_out/r4b/pmc_i1 -o i -- python3 bench.py --steps 60 --warmup 20 --repeats 1 --no-cpu-baseline --no-secondary > gpurun_out/r4b/pmc_i1.log 2>&1
python tools/pmc_summary.py $(find gpurun_out/r4b/pmc_i1 -name "*.db" | head -1) | grep "kernel\|---\|fused_chain_pc\|ens_gemm_pair" > gpurun_out/r4b/pmc_insts_1.md
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d gpurun_out/r4b/pmc_i2 -o i -- python3 bench.py --steps 60 --warmup 20 --repeats 1 --no-cpu-baseline --no-secondary > gpurun_out/r4b/pmc_i2.log 2>&1
python tools/pmc_summary.py $(find gpurun_out/r4b/pmc_i2 -name "*.db" | head -1) | grep "kernel\|---\|fused_chain_pc\|ens_gemm_pair" > gpurun_out/r4b/pmc_insts_2.md
cat gpurun_out/r4b/pmc_insts_1.md gpurun_out/r4b/pmc_insts_2.md; tail -2 gpurun_out/r4b/pmc_i2.log | cut -c1-200
rm -rf gpurun_out/r4b/pmc_i1 gpurun_out/r4b/pmc_i2
