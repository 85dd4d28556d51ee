#!/bin/bash
# Ablation copies of the library (tools/gpu_ablate.py): each skips one section of the MFCC frame
# pipeline while keeping its values live, so the section's share of the kernel time can be measured
# as a difference.  Tuning aid only; never loaded by the product.
set -e
cd "$(dirname "$0")/../mycroft_precise_amd/csrc"
mkdir -p build
for v in BASE TWIDDLE TRANSPOSE FFT EXCHANGE POWER MEL LOG DCT TOUCH PCM TABLES; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPE_ABL_$v -shared -o build/libpe_abl_$v.so engine.hip kernels.hip && echo built $v ) &
done
wait
