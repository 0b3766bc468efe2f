#!/bin/bash
# Build libssac_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# Each .hip is compiled to build/<name>.o (in parallel, only when it or a header changed), then linked.
#   ./build.sh          the product library
#   ./build.sh --lab    the LAB build (-DSSAC_LAB): the same library + the measurement scaffolding the tools under tools/
#                       need (s_memtime phase stamps, per-workgroup timelines; -DSSAC_EXPERIMENT_SKIP_* take effect).  It is
#                       a SEPARATE file, super_sac_amd/libssac_hip_lab.so (objects in build/obj_lab), loaded only when
#                       SSAC_LAB_BUILD=1 is in the environment (super_sac_amd/_lib.py): the product library is untouched.
set -e
cd "$(dirname "$0")"
SRC=super_sac_amd/csrc
OBJ=build/obj
OUT=super_sac_amd/libssac_hip.so
if [ "$1" = "--lab" ]; then shift; set -- -DSSAC_LAB "$@"; OBJ=build/obj_lab; OUT=super_sac_amd/libssac_hip_lab.so; fi
# --lab --tag NAME -D...: an experiment variant of the lab build, super_sac_amd/libssac_hip_lab_NAME.so (SSAC_LAB_TAG=NAME loads it)
if [ "$2" = "--tag" ]; then TAG=$3; first=$1; shift 3; set -- "$first" "$@"; OBJ=build/obj_lab_$TAG; OUT=super_sac_amd/libssac_hip_lab_$TAG.so; fi
mkdir -p "$OBJ"
# -ffp-contract=off: keep fp32 op boundaries as the reference's separate torch ops have them
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -I$SRC $*"
newest_hdr=$(ls -t include/*.h $SRC/*.h build.sh | head -1)
# a change of flags (extra arguments included) rebuilds everything: the flag string is kept beside the objects
if [ ! -f "$OBJ/.flags" ] || [ "$(cat "$OBJ/.flags")" != "$FLAGS" ]; then
    rm -f "$OBJ"/*.o
    printf '%s' "$FLAGS" > "$OBJ/.flags"
fi
pids=()
objs=()
for f in $SRC/*.hip; do
    o="$OBJ/$(basename "${f%.hip}").o"
    objs+=("$o")
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
        hipcc $FLAGS -c "$f" -o "$o" &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait "$p"; done
if [ ${#pids[@]} -gt 0 ] || [ ! -f "$OUT" ]; then
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
fi
