#!/bin/bash
# round 6: kept leftovers (pe_update_device_keep) + the prologue fixes of the record readers: new tests, then the headline and the
# capacity point with and without keep, the driver's 20-step command three times
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "kept or subset or host_fed or renumber or update_many" 2>&1 | tail -8 | tee $OUT/r6g_pytest_new.log
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('   %s: %.2f M windows/s, %.3f us/step, launch %.3f us, gru %.3f, mfcc %.3f' % (sys.argv[1].split('/')[-1], d['value'] / 1e6, 1e3 * d['ms_per_step'], 1e3 * d['roofline']['avg_launch_ms'],
      1e3 * d['roofline_gru']['avg_launch_ms'], 1e3 * d['roofline_mfcc']['avg_launch_ms']))
for e in d.get('extra_configs', []):
    print('      %s: %.1f M windows/s, update %.2f us, mfcc %.2f, network %.2f, parity %s' % (e['name'][:60], e['value'] / 1e6, 1e3 * e['stage_ms']['update_back_to_back'],
          1e3 * e['stage_ms']['mfcc_launch_alone'], 1e3 * e['stage_ms']['network_launch_alone'], e['parity']['ok']))
PY
}
for i in 1 2 3; do
  for k in 0 1; do
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --keep $k --no-cpu-baseline --no-extra-configs --no-batched > $OUT/r6g_driver_keep${k}_$i.json 2> $OUT/r6g_driver_keep${k}_$i.err
    show $OUT/r6g_driver_keep${k}_$i.json
  done
done
for k in 0 1; do
  timeout 600 python bench.py --keep $k --no-cpu-baseline --no-extra-configs --no-batched > $OUT/r6g_default_keep${k}.json 2> $OUT/r6g_default_keep${k}.err
  show $OUT/r6g_default_keep${k}.json
done
timeout 900 python bench.py --no-cpu-baseline --no-batched --only-extra capacity > $OUT/r6g_capacity.json 2> $OUT/r6g_capacity.err
show $OUT/r6g_capacity.json
timeout 900 python bench.py --no-cpu-baseline --no-batched --only-extra host-fed > $OUT/r6g_hostfed.json 2> $OUT/r6g_hostfed.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6g_hostfed.json'))
for e in d.get('extra_configs', []):
    print(json.dumps(e)[:700])
PY
