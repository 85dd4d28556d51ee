#!/bin/bash
# A/B on one box: the tree before the per-stream records (ab_old/, commit fd5314e) vs the current one -- the driver's 20-step command
# (five times each, interleaved) and the default command
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('   %s: %.2f M windows/s, %.3f us/step, launch %.3f us, gru %.3f, mfcc %.3f' % (sys.argv[1].split('/')[-1], d['value'] / 1e6, 1e3 * d['ms_per_step'], 1e3 * d['roofline']['avg_launch_ms'],
      1e3 * d['roofline_gru']['avg_launch_ms'], 1e3 * d['roofline_mfcc']['avg_launch_ms']))
PY
}
for i in 1 2 3 4 5; do
  for v in old new; do
    if [ $v = old ]; then B=$ROOT/ab_old/bench.py; else B=$ROOT/bench.py; fi
    (cd $(dirname $B) && timeout 600 python $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-batched > $OUT/r6c_driver_${v}_$i.json 2> $OUT/r6c_driver_${v}_$i.err)
    show $OUT/r6c_driver_${v}_$i.json
  done
done
for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then B=$ROOT/ab_old/bench.py; else B=$ROOT/bench.py; fi
    (cd $(dirname $B) && timeout 600 python $B --no-cpu-baseline --no-extra-configs --no-batched > $OUT/r6c_default_${v}_$i.json 2> $OUT/r6c_default_${v}_$i.err)
    show $OUT/r6c_default_${v}_$i.json
  done
done
