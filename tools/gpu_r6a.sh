#!/bin/bash
# round 6, first session: the new gates (self-launching bench, soaks, duplicate-grid switch, reserved engines on the XDL form) + the
# whole GPU suite + one default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "== new tests first"
timeout 1200 python -m pytest tests -m gpu -q -x -k "bench_starts_its_own or position_soak or general_float32_front_end_soak or colliding_mel_grid or large_launch or full_batch_update_many" 2>&1 | tail -15 | tee $OUT/r6a_pytest_new.log
echo "== whole suite"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/r6a_pytest_gpu.log
echo "== bench"
timeout 900 python bench.py > $OUT/r6a_bench.json 2> $OUT/r6a_bench.err; tail -3 $OUT/r6a_bench.err; cut -c1-600 $OUT/r6a_bench.json
python - <<'PY'
import json, os
d = json.load(open(os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r6a_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d.get('parity'))
print('time_batched', json.dumps(d.get('time_batched'), indent=1))
PY
