#!/bin/bash
mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_cases.py -m gpu -x -q -k "first_layer or pixel or drq or atari" > $O/small_test.log 2>&1; echo "tests exit $?" >> $O/small_test.log
tail -3 $O/small_test.log
python tools/first_wgrad_time.py atari 2>&1 | grep -v amdgpu
python tools/bench_pixels.py atari 40 2>&1 | tail -1
python -c "
import super_sac_amd.conv_encoder as ce
ce.FIRST_WGRAD_BANDS = False
import sys; sys.argv = ['bench_pixels.py', 'atari', '40']
exec(open('tools/bench_pixels.py').read())
" 2>&1 | tail -1
