#!/bin/bash
# PMC passes for the non-headline configurations (wide 256x2, bf16): matrix-core activity and HBM traffic.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {   # tag, bench args...
  tag=$1; shift
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    ptag=$(echo $pass | tr ' ' '_' | cut -c1-24)
    rm -rf $OUT/pmcx_${tag}_$ptag
    timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmcx_${tag}_$ptag -o pmc -- python $ROOT/bench.py --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmcx_${tag}_$ptag.err
    echo "== $tag / $pass rc=$?"
  done
}
run wide --units 256,256 --steps 12 --warmup 32
run bf16 --gru-precision bf16 --mfcc-precision f32 --steps 40 --warmup 40
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmcx_*/**/*counter_collection.csv', recursive=True):
    tag = f.split('/pmcx_')[1].split('_')[0]
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'pe::' not in k or 'clear' in k: continue
        short = tag + ':' + k.split('(')[0].replace('void ', '')
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/pmc_extra_summary.csv', 'w') as fo:
    fo.write('config:kernel,counter,dispatches,mean_per_dispatch\n')
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            line = '%s,%s,%d,%.6g' % (k, c, len(v), sum(v) / len(v))
            print(line); fo.write(line + '\n')
PY
