#!/bin/bash
# Experiment builds: libprecise_engine_<tag>.so under csrc/build/variants with extra -D flags.
#   tools/build_variants.sh tg16w2 "-DPE_TG=16 -DPE_WPE=2"  tg24w3 "-DPE_TG=24 -DPE_WPE=3" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/mycroft_precise_amd/csrc
mkdir -p $C/build/variants
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DPE_TUNING $flags -c $C/engine.hip -o $C/build/variants/engine_$tag.o &
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DPE_TUNING $flags -c $C/kernels.hip -o $C/build/variants/kernels_$tag.o &
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $C/build/variants/libprecise_engine_$tag.so $C/build/variants/engine_$tag.o $C/build/variants/kernels_$tag.o
    rm -f $C/build/variants/*_$tag.o; echo built $tag ) &
done
wait
