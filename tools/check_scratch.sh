#!/bin/bash
# tools/check_scratch.sh : device assembly of every .hip of the library; lists the kernels that use scratch (private
# segment) or spill registers.  The claim in DESIGN.md section 4 ("no kernel of the library uses scratch") is this output.
cd "$(dirname "$0")/.."
for f in super_sac_amd/csrc/*.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Isuper_sac_amd/csrc --cuda-device-only -S "$f" -o /tmp/cs_$$.s "$@" 2>/dev/null || { echo "$f: compile failed"; continue; }
  python3 - /tmp/cs_$$.s "$(basename $f)" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
n = bad = 0
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
    n += 1
    if int(m.group(2)) or int(m.group(4)):
        bad += 1
        print(f"  {sys.argv[2]}: {m.group(1)[:90]}: scratch {m.group(2)} B, {m.group(4)} spilled VGPRs ({m.group(3)} VGPRs)")
print(f"{sys.argv[2]}: {n} kernels, {bad} with scratch or spills")
PY
done
rm -f /tmp/cs_$$.s
