#!/bin/bash
# bench.py with the in-tree library and each variant library (tools/build_variants.sh): tools/gpu_lib_ab.sh <tag> "<bench args>" tag1 tag2 ...
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
tag=$1; args=$2; shift 2
for v in base "$@"; do
  if [ $v = base ]; then unset PE_LIB; else export PE_LIB=$PWD/mycroft_precise_amd/csrc/build/variants/libprecise_engine_$v.so; fi
  timeout 600 python bench.py --no-cpu-baseline --no-batched --no-extra-configs $args > $OUT/${tag}_$v.json 2> $OUT/${tag}_$v.err || tail -3 $OUT/${tag}_$v.err
  python3 - "$OUT/${tag}_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print('%-24s %7.1f M/s  step %.2f us  fused %.2f  mfcc %.2f  gru %.2f' % (sys.argv[2], d['value'] / 1e6, d['ms_per_step'] * 1e3,
          d['roofline']['avg_launch_ms'] * 1e3, d['roofline_mfcc']['avg_launch_ms'] * 1e3, d['roofline_gru']['avg_launch_ms'] * 1e3))
except Exception as ex:
    print(sys.argv[2], 'FAILED', ex)
PY
done
