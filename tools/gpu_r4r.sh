#!/bin/bash
# round 4: launch shape of the bf16 configuration at its per-GPU shard (8192 streams): frame workgroups per CU x network tiles per workgroup
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PE_LIB=$PWD/mycroft_precise_amd/csrc/build/variants/libprecise_engine_tun.so
: > $OUT/r4r_bf16_shape.log
for fw in 2 3 4; do for tpw in 1 2 4; do
  echo "== PE_FRAME_WG_PER_CU=$fw PE_BF16_TPW=$tpw" | tee -a $OUT/r4r_bf16_shape.log
  PE_FRAME_WG_PER_CU=$fw PE_BF16_TPW=$tpw python tools/gpu_sizes.py --mfcc f32 --gru bf16 --ring bf16 8192 2>&1 | grep streams | sed 's/mfcc alone.*//' | tee -a $OUT/r4r_bf16_shape.log
done; done
