#!/bin/bash
# tools/kasm.sh <file.hip> <kernel-name-substring> : device asm of one kernel -> /tmp/k.s (+ resource summary)
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Isuper_sac_amd/csrc --cuda-device-only -S "super_sac_amd/csrc/$1" -o /tmp/kasm_all.s $3 2>&1 | grep -E "error" || true
start=$(grep -n "^_Z.*$2.*:" /tmp/kasm_all.s | head -1 | cut -d: -f1)
end=$(awk -v s="$start" 'NR>s && /\.amdhsa_kernel/ {print NR; exit}' /tmp/kasm_all.s)
sed -n "${start},${end}p" /tmp/kasm_all.s > /tmp/k.s
grep -E "\.(vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count):" /tmp/kasm_all.s | head -0
awk -v s="$end" 'NR>=s && NR<s+60' /tmp/kasm_all.s | grep -E "next_free_vgpr|private_segment_fixed_size|accum_offset" 
wc -l /tmp/k.s
