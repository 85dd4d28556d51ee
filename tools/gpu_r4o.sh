#!/bin/bash
# round 4: frame workgroups per compute unit beside the faster critical-wave network role; roles alone (tuning library)
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PE_LIB=$PWD/mycroft_precise_amd/csrc/build/variants/libprecise_engine_tun.so
: > $OUT/r4o_frame_wg.log
for n in 2 3 4; do echo "== PE_FRAME_WG_PER_CU=$n" | tee -a $OUT/r4o_frame_wg.log; PE_FRAME_WG_PER_CU=$n python tools/gpu_sizes.py 4096 8192 2>&1 | grep streams | tee -a $OUT/r4o_frame_wg.log; done
for sk in 1 2; do echo "== PE_FUSED_SKIP=$sk (1: network role only, 2: MFCC roles only)" | tee -a $OUT/r4o_frame_wg.log; PE_FUSED_SKIP=$sk python tools/gpu_sizes.py 4096 2>&1 | grep streams | sed 's/mfcc alone.*//' | tee -a $OUT/r4o_frame_wg.log; done
unset PE_LIB
echo "== f32 front end tests after v_log_f32" | tee -a $OUT/r4o_frame_wg.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee -a $OUT/r4o_frame_wg.log
python tools/gpu_sizes.py --mfcc f32 --gru bf16 --ring bf16 8192 65536 2>&1 | grep streams | tee -a $OUT/r4o_frame_wg.log
