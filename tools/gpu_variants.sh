#!/bin/bash
# usage: tools/gpu_variants.sh "<python command>" tag1 tag2 ...   (runs the command with the in-tree library and each variant)
cmd=$1; shift
echo "== base"; $cmd 2>&1 | grep -v amdgpu.ids
for t in "$@"; do echo "== $t"; PE_LIB=$PWD/mycroft_precise_amd/csrc/build/variants/libprecise_engine_$t.so $cmd 2>&1 | grep -v amdgpu.ids; done
