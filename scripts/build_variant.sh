#!/bin/bash
# Builds a variant of the engine library for A/B measurements: scripts/build_variant.sh <name> [extra hipcc flags ...]
# -> build/variants/<name>.so (git-ignored; travels to the GPU box with gpurun; use with EMAT_LIB_PATH).  The shipped library
# stays delphy_amd/libemat_hip.so, built by csrc/Makefile alone.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/build/variants
cd $ROOT/delphy_amd/csrc
ID=$(cat emat_backend.hip emat_device_core.hpp emat_device_moves.hpp emat_device_spr.hpp emat_slab.hpp emat_gtree_kernels.hpp emat_build.hpp Makefile | sha256sum | cut -c1-16)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-unused-result -mllvm -amdgpu-lower-module-lds-strategy=module -DEMAT_BUILD_ID=\"$ID\" "$@" -shared -o $ROOT/build/variants/$NAME.so emat_backend.hip emat_run.cpp emat_dphy.cpp emat_multi.cpp -ldl
echo built build/variants/$NAME.so
