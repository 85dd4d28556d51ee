#!/bin/bash
# round 5, session c: new tests (host-fed pipeline, n_fft 16 / 32, bf16 layouts), bench with the host-fed extras, launch-shape sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
V=$ROOT/mycroft_precise_amd/csrc/build/variants
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $OUT/r5c_pytest_gpu.log
echo "== bench (driver command)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r5c_bench_driver.json 2> $OUT/r5c_bench_driver.err; tail -3 $OUT/r5c_bench_driver.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r5c_bench_driver.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])
for e in d.get('extra_configs', []):
    print(' ', e.get('name', '')[:70], '|', e.get('value'), e.get('ms_per_step'), e.get('parity', e.get('error', e.get('pcie', e.get('latency_us')))))
    for k in ('pageable_sources', 'synchronous_pe_update', 'bit_identical_to_device_resident_path'):
        if k in e: print('     ', k, e[k])
print('cpu', d.get('cpu_baseline', {}).get('value'))
PY
echo "== fused bf16 launch shape (tuning build): network tiles per workgroup x frames first"
for B in 8192 16384; do for tpw in 1 2 4; do for ff in 0 1; do
  echo -n "tpw=$tpw frames_first=$ff  "
  PE_BF16_TPW=$tpw PE_FUSED_FRAMES_FIRST=$ff PE_LIB=$V/libprecise_engine_r5t.so timeout 300 python tools/gpu_sizes.py --mfcc f32 --gru bf16 --ring bf16 $B 2>&1 | tail -1 | cut -c1-150
done; done; done | tee $OUT/r5c_bf16_launch_shape.log
du -sh $OUT | tail -1
