#!/bin/bash
# Debug copy of the library with MFCC section timers (-DPE_SECTION_TIMERS); never loaded by the product.
#   tools/build_debug.sh [tag [extra flags]]   ->  csrc/build/libprecise_engine_dbg[_tag].so   (PE_DBG_LIB selects one)
set -e
cd "$(dirname "$0")/../mycroft_precise_amd/csrc"
mkdir -p build
out=build/libprecise_engine_dbg${1:+_$1}.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DPE_SECTION_TIMERS $2 -shared -o $out engine.hip kernels.hip
echo built $out
