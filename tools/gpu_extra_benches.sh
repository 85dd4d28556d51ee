#!/bin/bash
# bench lines of the non-headline configurations (BASELINE configs[3], configs[4], large per-GPU batches)
#   tools/gpu_extra_benches.sh <tag>      -> gpurun_out/<tag>_bench_*.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; tag=$1; mkdir -p $OUT; cd $ROOT
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $OUT/${tag}_bench_$name.json 2> $OUT/${tag}_bench_$name.err; echo "== $name: $(cut -c1-330 $OUT/${tag}_bench_$name.json | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/value \1  ms_per_step \2/')"; }
run bf16_8192 --gru-precision bf16 --mfcc-precision f32 --ring-precision bf16 --streams 8192
run bf16_65536 --gru-precision bf16 --mfcc-precision f32 --ring-precision bf16 --streams 65536
run f64_65536 --streams 65536
run f64_8192 --streams 8192
run f32front_4096 --mfcc-precision f32
run wide256x2 --units 256,256
cd /tmp && export TMPDIR=/tmp
for cfg in "bf16_8192 --gru-precision bf16 --mfcc-precision f32 --ring-precision bf16 --streams 8192" "bf16_65536 --gru-precision bf16 --mfcc-precision f32 --ring-precision bf16 --streams 65536"; do
  set -- $cfg; name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/${tag}_pmc_${name}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${tag}_pmc_${name}_$c -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-batched --steps 40 --warmup 40 "$@" > /dev/null 2> $OUT/${tag}_pmc_${name}_$c.err
  done
done
python3 - "$tag" <<'PY'
import csv, glob, os, sys, collections
tag = sys.argv[1]
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out'
with open(out + '/%s_pmc_bf16_summary.csv' % tag, 'w') as fo:
    fo.write('config,kernel,counter,dispatches,mean_per_dispatch\n')
    for d in sorted(glob.glob(out + '/%s_pmc_bf16_*' % tag)):
        if not os.path.isdir(d): continue
        agg = collections.defaultdict(list)
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                k = r.get('Kernel_Name', '')
                if 'pe::' not in k or 'clear' in k: continue
                agg[(k.split('(')[0].replace('void ', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(agg.items()):
            line = '%s,%s,%s,%d,%.6g' % (os.path.basename(d).split('_pmc_')[1], k, c, len(v), sum(v) / len(v))
            print(line); fo.write(line + '\n')
PY
