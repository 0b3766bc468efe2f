#!/bin/bash
# same-box A/B of two SSAC_DEBUG settings on the full REDQ environment step (fp32 rows of tools/bench_configs.py)
for r in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export SSAC_DEBUG="$1"; else export SSAC_DEBUG="$2"; fi
    echo "$v $(python tools/bench_configs.py "REDQ (headline)" 2>/dev/null | grep "fp32" | cut -d'|' -f2,7 | tr '\n' ';')"
  done
done
