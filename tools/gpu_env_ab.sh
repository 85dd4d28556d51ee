#!/bin/bash
# bench.py under a list of environment settings, with the PE_TUNING variant library (tools/build_variants.sh tune ""):
#   tools/gpu_env_ab.sh <tag> "<bench args>" "ENV1=a ENV2=b" "ENV3=c" ...
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
tag=$1; args=$2; shift 2
export PE_LIB=$PWD/mycroft_precise_amd/csrc/build/variants/libprecise_engine_tune.so
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 600 python bench.py --no-cpu-baseline --no-batched $args > $OUT/${tag}_$i.json 2> $OUT/${tag}_$i.err || tail -3 $OUT/${tag}_$i.err
  python3 - "$OUT/${tag}_$i.json" "$e" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print('%-44s %7.1f M/s  step %.2f us  fused %.2f  mfcc %.2f  gru %.2f' % (sys.argv[2], d['value'] / 1e6, d['ms_per_step'] * 1e3,
          d['roofline']['avg_launch_ms'] * 1e3, d['roofline_mfcc']['avg_launch_ms'] * 1e3, d['roofline_gru']['avg_launch_ms'] * 1e3))
except Exception as ex:
    print(sys.argv[2], 'FAILED', ex)
PY
done
