#!/bin/bash
# Debug copy of the library with MFCC section timers (-DPE_SECTION_TIMERS); never loaded by the product.
set -e
cd "$(dirname "$0")/../mycroft_precise_amd/csrc"
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPE_SECTION_TIMERS -shared -o build/libprecise_engine_dbg.so engine.hip kernels.hip
echo built build/libprecise_engine_dbg.so
