#!/bin/bash
# round 6: per-step delivery (direct / host) at the headline's size; PMC of the MFCC launch at 65 536 streams with carried and
# with kept leftovers (WRITE_SIZE per stream, launch duration)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for m in direct host; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-batched --gather-every-step $m > $OUT/r6i_$m.json 2> $OUT/r6i_$m.err
  tail -2 $OUT/r6i_$m.err
  python - "$OUT/r6i_$m.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('%.1f M windows/s, %.3f us/step; per step: %s' % (d['value'] / 1e6, 1e3 * d['ms_per_step'], json.dumps(d['per_step_delivery'])[:400]))
PY
done
timeout 600 python -m pytest tests -m gpu -q -x -k "delivers_per_step" 2>&1 | tail -3
for how in update keep; do
  bash tools/gpu_pmc_any.sh mfcc65536_$how "python tools/gpu_mfcc_only.py 65536 60 f64 $how" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" > $OUT/r6i_pmc_$how.log 2>&1
  grep "mfcc_kernel" $OUT/pmcany_mfcc65536_${how}_summary.csv
  cd /tmp && export TMPDIR=/tmp
  rm -rf $OUT/r6i_trace_$how
  (cd $ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6i_trace_$how -o t -- python tools/gpu_mfcc_only.py 65536 200 f64 $how > /dev/null 2> $OUT/r6i_trace_$how.err)
  for f in $(find $OUT/r6i_trace_$how -name "*kernel_stats.csv"); do grep "pe::" $f | cut -c1-200 | tee $OUT/r6i_kernel_stats_$how.csv; done
  rm -rf $OUT/r6i_trace_$how $OUT/pmcany_mfcc65536_${how}_[0-9]
  cd $ROOT
done
