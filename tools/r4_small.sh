#!/bin/bash
mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "first_layer" > $O/small_test.log 2>&1; echo "tests exit $?" >> $O/small_test.log
tail -3 $O/small_test.log
python tools/first_wgrad_time.py dmc 2>&1 | grep -v amdgpu
python tools/first_wgrad_time.py atari 2>&1 | grep -v amdgpu
python tools/bench_pixels.py dmc 40 2>&1 | tail -1
python tools/bench_pixels.py atari 40 2>&1 | tail -1
