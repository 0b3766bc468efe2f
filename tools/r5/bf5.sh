#!/bin/bash
O=gpurun_out/r5/bf5; mkdir -p $O
timeout 300 python tools/r5/rc_phases.py 65536 > $O/phases_65536.txt 2>&1
timeout 300 python tools/r5/rc_phases.py 4096 > $O/phases_4096.txt 2>&1
tail -3 $O/phases_65536.txt; tail -3 $O/phases_4096.txt
