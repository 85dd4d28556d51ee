#!/bin/bash
# usage: tools/gpu_pmc_any.sh <tag> "<python command relative to repo root>" "<counters pass 1>" ["<counters pass 2>" ...]
# One rocprofv3 --pmc run per counter list (kernel trace only), then mean per dispatch and kernel.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
tag=$1; cmd=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  rm -rf $OUT/pmcany_${tag}_$i
  ( cd $ROOT && timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmcany_${tag}_$i -o pmc -- $cmd > /dev/null 2> $OUT/pmcany_${tag}_$i.err )
  echo "== pass $i ($pass) rc=$?"
done
python3 - "$tag" <<'PY'
import csv, glob, os, sys, collections
tag = sys.argv[1]
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmcany_%s_*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'pe::' not in k or 'clear' in k: continue
        agg[k.split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/pmcany_%s_summary.csv' % tag, 'w') as fo:
    fo.write('kernel,counter,dispatches,mean_per_dispatch\n')
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            line = '%s,%s,%d,%.6g' % (k, c, len(v), sum(v) / len(v))
            print(line); fo.write(line + '\n')
PY
