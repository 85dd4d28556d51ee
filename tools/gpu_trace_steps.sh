#!/bin/bash
# Kernel-trace timeline of the driver's command (bench.py --steps 20 --warmup 5): per-dispatch start/end of the fused launches.
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace20
# (optional 3rd argument: extra bench.py flags; PE_LIB / PE_FUSED_SKIP ... are inherited from the environment)
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace20 -o t -- python $R/bench.py --steps ${1:-20} --warmup ${2:-5} --no-cpu-baseline --no-extra-configs --no-batched ${3:-} > $OUT/trace20.json 2> $OUT/trace20.err
python3 - $OUT <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
d = json.load(open(out + '/trace20.json'))
print('bench line: %.1f M/s, %.2f us/step' % (d['value'] / 1e6, d['ms_per_step'] * 1e3))
f = glob.glob(out + '/trace20/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'fused_update' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
prev_end = None
import os
n_show = int(os.environ.get('TRACE_SHOW', '45'))
n_from = int(os.environ.get('TRACE_FROM', '0'))
for i, r in enumerate(rows[n_from:n_from + n_show]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%3d start %9.2f us  dur %6.2f  gap %6.2f' % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0))
    prev_end = e
durs = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[5:])
print('durations of %d launches: min %.2f  median %.2f  p75 %.2f  max %.2f  mean %.2f' % (len(durs), durs[0], durs[len(durs) // 2], durs[3 * len(durs) // 4], durs[-1], sum(durs) / len(durs)))
PY
