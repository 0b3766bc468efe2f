#!/bin/bash
python tools/first_wgrad_time.py dmc 2>&1 | grep -v amdgpu
