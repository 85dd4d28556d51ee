#!/bin/bash
# A/B of bench.py variants on one box: tools/gpu_ab.sh <tag> "<args A>" "<args B>" ...   -> gpurun_out/<tag>_<i>.json
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
tag=$1; shift
i=0
for a in "$@"; do
  i=$((i+1))
  timeout 600 python bench.py --no-cpu-baseline $a > $OUT/${tag}_$i.json 2> $OUT/${tag}_$i.err || tail -3 $OUT/${tag}_$i.err
  python3 - "$OUT/${tag}_$i.json" "$a" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    tb = d.get('time_batched') or {}
    print('%-52s %7.1f M/s  step %.2f us  fused %.2f  mfcc %.2f  gru %.2f  many8 %s' % (sys.argv[2], d['value'] / 1e6, d['ms_per_step'] * 1e3,
          d['roofline']['avg_launch_ms'] * 1e3, d['roofline_mfcc']['avg_launch_ms'] * 1e3, d['roofline_gru']['avg_launch_ms'] * 1e3,
          '%.1f M/s' % (tb['value'] / 1e6) if tb else '-'))
except Exception as ex:
    print(sys.argv[2], 'FAILED', ex)
PY
done
