#!/bin/bash
# The large-batch points of the front end and of the fused launch: bench.py at 8192 / 65536 streams, float64 front end + f32
# network and float32 front end + bf16 network (BASELINE configs[4]).   tools/gpu_scale_points.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
tag=${1:-sc}
for cfg in "f64_8192 --streams 8192" "f64_65536 --streams 65536 --steps 100" \
           "bf16_8192 --streams 8192 --mfcc-precision f32 --gru-precision bf16 --ring-precision bf16" \
           "bf16_65536 --streams 65536 --steps 100 --mfcc-precision f32 --gru-precision bf16 --ring-precision bf16"; do
  set -- $cfg; name=$1; shift
  timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --no-batched "$@" > $OUT/${tag}_$name.json 2> $OUT/${tag}_$name.err || tail -3 $OUT/${tag}_$name.err
  python3 - "$OUT/${tag}_$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('%-12s %7.1f M/s  step %7.2f us  fused %7.2f (%s %.3f)  mfcc %7.2f (hbm %.3f)  gru %7.2f (mfma %.3f)' % (sys.argv[2], d['value'] / 1e6, d['ms_per_step'] * 1e3,
      d['roofline']['avg_launch_ms'] * 1e3, d['roofline']['bound'], d['roofline']['frac'], d['roofline_mfcc']['avg_launch_ms'] * 1e3, d['roofline_mfcc']['frac'],
      d['roofline_gru']['avg_launch_ms'] * 1e3, d['roofline_gru']['frac']))
PY
done
