#!/bin/bash
# (round 5) quad MFCC: timing-only ablations of the leftover role (a: none, b: loads without stores) next to the full kernel
V=mycroft_precise_amd/csrc/build/variants
for rep in 1 2; do
for lib in quad4 quad4a quad4b; do
  echo "== $lib"
  PE_QUAD=1 PE_QUAD_WG_PER_CU=2 PE_LIB=$PWD/$V/libprecise_engine_$lib.so timeout 300 python tools/gpu_quad_check.py 65536 f64 2>&1 | grep "MFCC launch"
done
done
