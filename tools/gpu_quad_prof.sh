#!/bin/bash
# (round 5) quad MFCC: kernel trace + SQ counters of the quad frames kernel next to the one-frame-per-wave kernel
V=mycroft_precise_amd/csrc/build/variants
lib=${1:-quad4}; per=${2:-2}; prec=${3:-f64}
export PE_LIB=$PWD/$V/libprecise_engine_$lib.so PE_QUAD_WG_PER_CU=$per
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for q in 0 1; do
  rm -rf $ROOT/gpurun_out/quadtrace_$q
  ( cd $ROOT && PE_QUAD=$q timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/quadtrace_$q -o t -- python tools/gpu_quad_check.py 65536 $prec > /dev/null 2>&1 )
  echo "== PE_QUAD=$q kernel stats"; find $ROOT/gpurun_out/quadtrace_$q -name '*kernel_stats.csv' | head -1 | xargs -r head -8 | cut -c1-200
done
cd $ROOT
PE_QUAD=1 bash tools/gpu_pmc_any.sh quad$lib "python tools/gpu_quad_check.py 65536 $prec" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" 2>&1 | grep -v "clear\|gru"
