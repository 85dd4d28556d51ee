#!/bin/bash
# where the general front end's time goes: ablation builds (tools/build_variants.sh gablN "-DPE_GEN_ABL=N")
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
tag=${1:-r4h}
: > $OUT/${tag}_general.log
for v in product "$@"; do
  [ "$v" = "$tag" ] && continue
  if [ $v = product ]; then unset PE_LIB; else export PE_LIB=$PWD/mycroft_precise_amd/csrc/build/variants/libprecise_engine_$v.so; fi
  timeout 300 python tools/gpu_general.py 2>&1 | grep streams | tee -a $OUT/${tag}_general.log
done
unset PE_LIB
timeout 300 python tools/gpu_general.py --n-fft 512 --n-filt 20 --n-mfcc 13 2>&1 | grep streams | tee -a $OUT/${tag}_general.log
timeout 300 python tools/gpu_general.py --n-fft 2048 --n-filt 80 --n-mfcc 32 2>&1 | grep streams | tee -a $OUT/${tag}_general.log
timeout 300 python tools/gpu_general.py --mfcc f32 2>&1 | grep streams | tee -a $OUT/${tag}_general.log
