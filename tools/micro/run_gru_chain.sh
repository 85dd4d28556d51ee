#!/bin/bash
# GPU-box driver of tools/micro/gru_chain.hip: timers build + plain build, then a rocprofv3 kernel trace of the plain one.
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=tools/micro/build
echo "== timers build"; timeout 40 $B/gru_chain_t ${1:-4096} 2>&1 | tee $OUT/gru_chain_t.log
echo "== plain build"; timeout 40 $B/gru_chain ${1:-4096} 2>&1 | tee $OUT/gru_chain.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/gru_chain_prof
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gru_chain_prof -o t -- $OLDPWD/$B/gru_chain ${1:-4096} > /dev/null 2>&1
for f in $(find $OUT/gru_chain_prof -name "*kernel_stats.csv"); do cut -c1-160 $f | tee $OUT/gru_chain_kernel_stats.csv; done
