#!/bin/bash
# PMC passes (own runs, kernel-trace only): HBM traffic and matrix-core / LDS activity of the pe:: kernels.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --steps 40 --warmup 40"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf $OUT/pmc_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -o pmc -- python $ROOT/bench.py $ARGS > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
  echo "== $pass: rc=$?"; tail -2 $OUT/pmc_$tag.err | cut -c1-200
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'pe::' not in k: continue
        short = k.split('(')[0].replace('void ', '')
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/pmc_summary.csv', 'w') as fo:
    fo.write('kernel,counter,dispatches,mean_per_dispatch\n')
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            line = '%s,%s,%d,%.6g' % (k, c, len(v), sum(v) / len(v))
            print(line); fo.write(line + '\n')
PY
