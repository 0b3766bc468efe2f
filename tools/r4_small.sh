#!/bin/bash
mkdir -p gpurun_out/r4c
for i in 1 2; do timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2; done
