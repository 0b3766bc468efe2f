cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_cases.py -x -q -m gpu -k "polyak or pixel or drq or atari or conv" > gpurun_out/r4b/pix_tests_$1.log 2>&1; tail -3 gpurun_out/r4b/pix_tests_$1.log
for c in dmc atari; do timeout 300 python tools/bench_pixels.py $c 60 2>&1 | tail -2; done | tee gpurun_out/r4b/pix_rows_$1.log
