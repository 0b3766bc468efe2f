#!/bin/bash
mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_cases.py -m gpu -x -q -k "streamed or pixel or drq or atari or encoder" > $O/small_test.log 2>&1; echo "tests exit $?" >> $O/small_test.log
tail -3 $O/small_test.log
python tools/bench_pixels.py dmc 40 2>&1 | tail -1
python tools/bench_pixels.py atari 40 2>&1 | tail -1
