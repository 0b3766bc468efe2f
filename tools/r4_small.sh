#!/bin/bash
mkdir -p gpurun_out/r4c
timeout 1200 python -m pytest tests/test_hip_cases.py -m gpu -x -q -k "full_size_pixel" 2>&1 | tail -3
