#!/bin/bash
# round 6, second session: per-stream state records + pe_update_subset
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "== subset tests"
timeout 1200 python -m pytest tests -m gpu -q -x -s -k "advance_independently or subset or renumbered" 2>&1 | tail -25 | tee $OUT/r6b_pytest_subset.log
echo "== whole suite"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/r6b_pytest_gpu.log
echo "== bench"
timeout 900 python bench.py > $OUT/r6b_bench.json 2> $OUT/r6b_bench.err; tail -3 $OUT/r6b_bench.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r6b_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'parity', d.get('parity', {}).get('max_abs_err'))
print('gru', d['roofline_gru']['avg_launch_ms'], 'mfcc', d['roofline_mfcc']['avg_launch_ms'])
print('time_batched', d['time_batched']['value'], d['time_batched']['other_form']['value'])
for x in d.get('extra_configs', []):
    print(x.get('name', '')[:60], x.get('value'), x.get('error'), (x.get('parity') or {}).get('ok'))
PY
echo "== bench, the driver's command"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-configs > $OUT/r6b_bench_driver.json 2> $OUT/r6b_bench_driver.err; cut -c1-200 $OUT/r6b_bench_driver.json
