#!/bin/bash
# round 5, session b: the product with the five-values bf16 network + unpacked float32 frame kernels: parity suite, 1e8-frame soaks,
# launch-shape sweep of the fused bf16 update at 8192 streams (tuning build of the same source), timings.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
V=$ROOT/mycroft_precise_amd/csrc/build/variants
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $OUT/r5b_pytest_gpu.log
echo "== soak 1e8 float32 frames (product library)"
( timeout 900 python tools/gpu_frame_soak.py --streams 8192 --frames 1e8 --ring bf16 2>&1 | tail -1
  timeout 900 python tools/gpu_frame_soak.py --streams 65536 --frames 1e8 --ring bf16 2>&1 | tail -1
  timeout 900 python tools/gpu_frame_soak.py --streams 8192 --frames 3e7 --ring f32 2>&1 | tail -1
  timeout 900 python tools/gpu_frame_soak.py --streams 4096 --frames 3e7 --gru f32 2>&1 | tail -1 ) | tee $OUT/r5b_soak.log
echo "== time (product)"
( timeout 300 python tools/gpu_sizes.py --mfcc f32 --gru bf16 --ring bf16 8192 16384 65536 2>&1 | grep streams
  timeout 300 python tools/gpu_sizes.py 4096 2>&1 | grep streams ) | tee $OUT/r5b_time.log
echo "== fused bf16 launch shape at 8192 / 16384 streams (tuning build): network tiles per workgroup x frames first"
for B in 8192 16384; do for tpw in 1 2 4; do for ff in 0 1; do
  echo -n "tpw=$tpw frames_first=$ff  "
  PE_BF16_TPW=$tpw PE_FUSED_FRAMES_FIRST=$ff PE_LIB=$V/libprecise_engine_r5t.so timeout 300 python tools/gpu_sizes.py --mfcc f32 --gru bf16 --ring bf16 $B 2>&1 | grep streams | cut -c1-120
done; done; done | tee $OUT/r5b_bf16_launch_shape.log
du -sh $OUT | tail -1
