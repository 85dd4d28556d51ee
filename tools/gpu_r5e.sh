#!/bin/bash
# round 5, session e: where the time of the wide XDL kernel goes: ablation (no weight loads) + PMC of both forms
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
V=$ROOT/mycroft_precise_amd/csrc/build/variants
( PE_WIDE_TILING=2 python tools/gpu_wide.py 256,256 4096 2>&1 | tail -1
  PE_WIDE_TILING=2 PE_LIB=$V/libprecise_engine_wx3a.so python tools/gpu_wide.py 256,256 4096 2>&1 | tail -1 ) | tee $OUT/r5e_wide_ablation.log
for t in 2 0; do
PE_WIDE_TILING=$t tools/gpu_pmc_any.sh r5e_wide$t "python tools/gpu_wide.py 256,256 4096" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
  "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" 2>&1 | grep "gru_wide"
done
rm -rf $OUT/pmcany_r5e_wide*_[0-9]
