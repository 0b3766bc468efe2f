#!/bin/bash
mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_cases.py -m gpu -x -q -k "shift or aug or first_layer or pixel or drq" > $O/small_test.log 2>&1; echo "tests exit $?" >> $O/small_test.log
tail -3 $O/small_test.log
python tools/first_shift_time.py dmc 2>&1 | grep "two launches"
python tools/first_shift_time.py atari 2>&1 | grep "two launches"
python tools/bench_pixels.py dmc 40 2>&1 | tail -1
python tools/bench_pixels.py atari 40 2>&1 | tail -1
