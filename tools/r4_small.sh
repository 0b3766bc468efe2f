#!/bin/bash
mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_cases.py -m gpu -x -q -k "conv or pixel or drq or atari or first_layer" > $O/small_test.log 2>&1; echo "tests exit $?" >> $O/small_test.log
tail -3 $O/small_test.log
python tools/conv_wgrad_time.py 1024 20 32 64 4 2 384 2>&1 | grep -v amdgpu
python tools/conv_wgrad_time.py 512 41 32 32 3 1 3328 2>&1 | grep -v amdgpu
python tools/bench_pixels.py dmc 40 2>&1 | tail -1
python tools/bench_pixels.py atari 40 2>&1 | tail -1
