#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace.  Run via gpurun.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "== pytest -m gpu" 
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; cat $OUT/bench.json
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_prof.json 2> $OUT/rocprof.err
tail -3 $OUT/rocprof.err
find $OUT/prof -name "*stats*" | head
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -12 $f; done
