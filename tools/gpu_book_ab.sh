#!/bin/bash
# (round 5) bookkeeping role with 16-byte moves: MFCC launch alone, shipped library vs a library built before the change (A/B on one box)
OLD=$PWD/mycroft_precise_amd/csrc/build/variants/libprecise_engine_${1:-quad4}.so
for rep in 1 2; do
for prec in f64 f32; do
  for n in 65536 8192; do
    echo "== $prec $n streams: before | after"
    PE_LIB=$OLD timeout 300 python tools/gpu_quad_check.py $n $prec 2>&1 | grep "MFCC launch"
    timeout 300 python tools/gpu_quad_check.py $n $prec 2>&1 | grep "MFCC launch\|oracle"
  done
done
done
