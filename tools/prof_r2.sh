cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o t -- python3 bench.py --steps 400 --warmup 100 --repeats 3 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o f -- python3 bench.py --steps 60 --warmup 20 --repeats 1 --no-cpu-baseline --no-secondary > $O/pf.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o w -- python3 bench.py --steps 60 --warmup 20 --repeats 1 --no-cpu-baseline --no-secondary > $O/pw.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pm -o m -- python3 bench.py --steps 60 --warmup 20 --repeats 1 --no-cpu-baseline --no-secondary > $O/pm.log 2>&1
python tools/rocpd_summary.py $O/kt/t_results.db | head -12
ls $O/*
timeout 200 python bench.py --steps 20 --warmup 5 > $O/b20.json 2> $O/b20.err; tail -c 600 $O/b20.json
