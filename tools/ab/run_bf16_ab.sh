#!/bin/bash
# same-box A/B of two SSAC_DEBUG settings, bf16 rows of tools/bench_configs.py      tools/ab/run_bf16_ab.sh "<A>" "<B>"
for r in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export SSAC_DEBUG="$1"; else export SSAC_DEBUG="$2"; fi
    echo "== $v ($SSAC_DEBUG)"
    python tools/bench_configs.py "REDQ" 2>/dev/null | grep "bf16" | grep -v "env step" | cut -d'|' -f2,5,6,7
  done
done
