#!/bin/bash
# Build libssac_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
# -ffp-contract=off: keep fp32 op boundaries as the reference's separate torch ops have them
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -Iinclude -Isuper_sac_amd/csrc \
    -o super_sac_amd/libssac_hip.so super_sac_amd/csrc/*.hip "$@"
