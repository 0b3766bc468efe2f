#!/bin/bash
# same-box A/B of two SSAC_DEBUG settings on the pixel configurations     tools/ab/run_pix_ab.sh "<A>" "<B>"
for r in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export SSAC_DEBUG="$1"; else export SSAC_DEBUG="$2"; fi
    for c in dmc atari; do echo "$v $c $(python tools/bench_pixels.py $c 20 2>/dev/null | tail -1)"; done
  done
done
