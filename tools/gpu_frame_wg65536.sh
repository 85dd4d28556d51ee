#!/bin/bash
# (round 5) MFCC launch alone at 65 536 streams: resident frame workgroups per CU (PE_FRAME_WG_PER_CU, tuning build), same box
V=mycroft_precise_amd/csrc/build/variants
for rep in 1 2; do
for prec in f64 f32; do
for per in 4 3 2 5; do
  echo "== $prec, $per frame workgroups per CU"
  PE_FRAME_WG_PER_CU=$per PE_LIB=$PWD/$V/libprecise_engine_tune.so timeout 300 python tools/gpu_quad_check.py 65536 $prec 2>&1 | grep "MFCC launch"
done
done
done
