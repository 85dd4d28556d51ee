#!/bin/bash
# round 5, session a: the float32-frame x XDL question (reproducer, packed vs unpacked frame role with the five-values
# network: wrong streams, soak, time) + a fresh PMC of the float64 MFCC launch at 65 536 streams.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
V=$ROOT/mycroft_precise_amd/csrc/build/variants
echo "== reproducer"
timeout 300 tools/micro/build/pk_swap_hazard > $OUT/r5a_pk_swap_hazard.csv 2>&1; cat $OUT/r5a_pk_swap_hazard.csv
echo "== wrong streams, five-values network, packed / unpacked frame role"
for lib in b20p b20u; do for B in 8192 65536; do
  PE_B20=1 PE_LIB=$V/libprecise_engine_$lib.so timeout 300 python tools/gpu_b20_debug.py $B 2>&1 | tail -1
done; done | tee $OUT/r5a_b20_wrong_streams.log
echo "== soak (2e7 frames each), unpacked frame role + five-values network"
for B in 8192 65536; do
  PE_B20=1 PE_LIB=$V/libprecise_engine_b20u.so timeout 600 python tools/gpu_frame_soak.py --streams $B --frames 2e7 2>&1 | tail -1
done | tee $OUT/r5a_soak_b20u.log
echo "== time"
( timeout 300 python tools/gpu_sizes.py --mfcc f32 --gru bf16 --ring bf16 8192 65536 2>&1 | sed 's/^/product          /'
  PE_B20=1 PE_LIB=$V/libprecise_engine_b20p.so timeout 300 python tools/gpu_sizes.py --mfcc f32 --gru bf16 --ring bf16 8192 65536 2>&1 | sed 's/^/b20 packed       /'
  PE_B20=1 PE_LIB=$V/libprecise_engine_b20u.so timeout 300 python tools/gpu_sizes.py --mfcc f32 --gru bf16 --ring bf16 8192 65536 2>&1 | sed 's/^/b20 unpacked     /'
  PE_B20=0 PE_LIB=$V/libprecise_engine_b20u.so timeout 300 python tools/gpu_sizes.py --mfcc f32 --gru bf16 --ring bf16 8192 65536 2>&1 | sed 's/^/8-values unpacked /'
  PE_LIB=$V/libprecise_engine_b20u.so timeout 300 python tools/gpu_sizes.py --mfcc f32 --gru f32 4096 65536 2>&1 | sed 's/^/f32net unpacked   /'
  timeout 300 python tools/gpu_sizes.py --mfcc f32 --gru f32 4096 65536 2>&1 | sed 's/^/f32net product    /'
) | tee $OUT/r5a_time_packed_unpacked.log
echo "== PMC, float64 MFCC launch at 65536 streams"
tools/gpu_pmc_any.sh r5a_mfcc65536 "python tools/gpu_mfcc_only.py 65536 40 f64" "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" 2>&1 | tail -40
rm -rf $OUT/pmcany_r5a_mfcc65536_[0-9]
du -sh $OUT | tail -1
