#!/bin/bash
# (round 4) the x3 network launch alone at 65536 / 4096 streams: microbenchmark rows for the XDL pipe, then PMC passes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 120 tools/micro/build/pipe_overlap x 2>&1 | tail -8 > gpurun_out/r4x_pipe_overlap_xdl_operands.csv
cat gpurun_out/r4x_pipe_overlap_xdl_operands.csv
export PE_GRU_TILING=2
bash tools/gpu_pmc_any.sh x3net "python tools/gpu_gru_only.py 65536 30" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES" 2>&1 | grep -v "^==" | grep "x3"
