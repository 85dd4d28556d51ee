#!/bin/bash
# (round 5) quad MFCC: kernel trace of the 65536-stream MFCC launch, one frame per wave vs four (frames kernel + book kernel)
V=mycroft_precise_amd/csrc/build/variants
lib=${1:-quad4}; per=${2:-2}
export PE_LIB=$PWD/$V/libprecise_engine_$lib.so PE_QUAD_WG_PER_CU=$per
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for prec in f64 f32; do
for q in 0 1; do
  rm -rf $ROOT/gpurun_out/quadtrace_$q
  ( cd $ROOT && PE_QUAD=$q timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/quadtrace_$q -o t -- python tools/gpu_quad_check.py 65536 $prec > /dev/null 2>&1 )
  python3 - $ROOT/gpurun_out/quadtrace_$q $q $prec <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mfcc' in k:
            d[k.split('(')[0].replace('void pe::', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    big = sorted(v)[len(v) // 2:]          # the 65536-stream launches (the parity part of the tool runs 37 streams)
    print('PE_QUAD=%s %s  %-60s launches %d, median of the 65536-stream half %.2f us' % (sys.argv[2], sys.argv[3], k, len(v), big[len(big) // 2]))
PY
done
done
