#!/bin/bash
# One GPU-box session for the committed evidence: parity tests, smoke, bench, rocprofv3 kernel trace, PMC passes.
#   tools/gpu_round.sh <tag> [bench args...]      -> gpurun_out/<tag>_*
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
tag=$1; shift
mkdir -p $OUT
cd $ROOT
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/${tag}_pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/${tag}_smoke.log
echo "== bench"
timeout 900 python bench.py "$@" > $OUT/${tag}_bench.json 2> $OUT/${tag}_bench.err; tail -2 $OUT/${tag}_bench.err; cut -c1-400 $OUT/${tag}_bench.json
echo "== bench, the driver's command"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${tag}_bench_driver.json 2> $OUT/${tag}_bench_driver.err; tail -2 $OUT/${tag}_bench_driver.err; cut -c1-300 $OUT/${tag}_bench_driver.json
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${tag}_prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_prof -o trace -- python $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/${tag}_bench_prof.json 2> $OUT/${tag}_rocprof.err
for f in $(find $OUT/${tag}_prof -name "*kernel_stats.csv"); do
  python3 - "$f" > $OUT/${tag}_kernel_stats.csv <<'PY'
import csv, re, sys
rows = list(csv.reader(open(sys.argv[1])))
print(','.join(rows[0]))
for r in rows[1:]:
    if 'pe::' not in r[0]:
        continue
    name = re.sub(r'\(pe::.*$', '', r[0]).replace('void ', '')        # kernel name with its template arguments, without the parameter list
    print(','.join(['"%s"' % name] + r[1:]))
PY
  cat $OUT/${tag}_kernel_stats.csv
done
echo "== PMC passes"
ARGS="--no-cpu-baseline --no-batched --no-extra-configs --steps 40 --warmup 40 $*"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf $OUT/${tag}_pmc_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/${tag}_pmc_$i -o pmc -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${tag}_pmc_$i.err
  echo "== pass $i ($pass): rc=$?"
done
python3 - "$tag" "$@" <<'PY'
import csv, glob, os, sys, collections
tag = sys.argv[1]
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/%s_pmc_*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'pe::' not in k or 'clear' in k: continue
        agg[k.split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
import json
traffic = {'source': 'tools/gpu_round.sh %s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py --no-batched --steps 40, mean per dispatch' % tag,
           'note': 'bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, raw counters; see profiles/README.md for the calibration of this access pattern'}
for k in agg:
    if 'FETCH_SIZE' in agg[k] and 'WRITE_SIZE' in agg[k]:
        f, w = agg[k]['FETCH_SIZE'], agg[k]['WRITE_SIZE']
        traffic[k.split('<')[0].replace('pe::', '')] = {'fetch_kb': round(sum(f) / len(f), 1), 'write_kb': round(sum(w) / len(w), 1), 'kernel': k}
rest = sys.argv[2:]
traffic['streams'] = int(rest[rest.index('--streams') + 1]) if '--streams' in rest else 4096
json.dump(traffic, open(out + '/%s_pmc_latest.json' % tag, 'w'), indent=1)
with open(out + '/%s_pmc_summary.csv' % tag, 'w') as fo:
    fo.write('kernel,counter,dispatches,mean_per_dispatch\n')
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            line = '%s,%s,%d,%.6g' % (k, c, len(v), sum(v) / len(v))
            print(line); fo.write(line + '\n')
PY
# gpurun merges at most 64 MiB back: keep the summaries and ONE raw kernel trace (gzip), drop the counter dumps
cd $OUT
for f in $(find ${tag}_prof -name "*kernel_trace.csv" 2>/dev/null); do gzip -c "$f" > ${tag}_kernel_trace.csv.gz; done
rm -rf ${tag}_prof ${tag}_pmc_[0-9]
du -sh $OUT | tail -1
