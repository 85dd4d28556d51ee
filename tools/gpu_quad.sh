#!/bin/bash
# (round 5) four-frames-per-wave MFCC experiment: parity + MFCC launch time per variant library, same box
#   tools/gpu_quad.sh lib[:wg_per_cu_f64[:wg_per_cu_f32]] ...
V=mycroft_precise_amd/csrc/build/variants
first=1
for spec in "$@"; do
  IFS=: read lib p64 p32 <<< "$spec"
  for prec in f64 f32; do
    per=${p64:-1}; [ $prec = f32 ] && per=${p32:-${p64:-1}}
    if [ $first = 1 ]; then
      echo "== one frame per wave, $prec"
      PE_QUAD=0 PE_LIB=$PWD/$V/libprecise_engine_$lib.so timeout 300 python tools/gpu_quad_check.py 65536 $prec 2>&1 | grep -v amdgpu.ids
    fi
    echo "== $lib $prec, $per workgroups per CU"
    PE_QUAD=1 PE_QUAD_WG_PER_CU=$per PE_LIB=$PWD/$V/libprecise_engine_$lib.so timeout 300 python tools/gpu_quad_check.py 65536 $prec 2>&1 | grep -v amdgpu.ids
  done
  first=0
done
