#!/bin/bash
# Round 4: the throughput regime.  (1) which pipes of a SIMD overlap across waves (tools/micro/pipe_overlap), (2) the new
# parity tests, (3) fused / MFCC-alone / network-alone times at large batches for the launch shapes: one tile per wave
# (round 3 policy) vs two tiles per wave + mixed workgroups, each role of the mixed launch alone (PE_FUSED_SKIP), one or
# two mixed workgroups per compute unit.  Needs the tuning library:  tools/build_variants.sh tun ""
#   tools/gpu_mix.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
tag=${1:-r4b}
echo "== pipe overlap"; timeout 120 tools/micro/build/pipe_overlap 2>&1 | tee $OUT/${tag}_pipe_overlap.csv
echo "== new parity tests (product library)"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "two_tiles or capacity_batch or use_delta or kernel_shapes" 2>&1 | tail -5 | tee $OUT/${tag}_pytest_new.log
export PE_LIB=$PWD/mycroft_precise_amd/csrc/build/variants/libprecise_engine_tun.so
run() {  # label, env..., -- args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "-- $label: ${envs[*]} $*"
  env "${envs[@]}" timeout 300 python tools/gpu_sizes.py "$@" 2>&1 | grep streams | sed "s/^/[$label] /" | tee -a $OUT/${tag}_sizes.log
}
: > $OUT/${tag}_sizes.log
for prec in f64 f32; do
  run "one-tile-$prec"        PE_PAIR=0 -- --mfcc $prec 65536 32768
  run "mix-$prec"             PE_PAIR=1 -- --mfcc $prec 65536 32768 16384
  run "mix-net-only-$prec"    PE_PAIR=1 PE_FUSED_SKIP=1 -- --mfcc $prec 65536
  run "mix-frames-only-$prec" PE_PAIR=1 PE_FUSED_SKIP=2 -- --mfcc $prec 65536
  run "mix-1wg-$prec"         PE_PAIR=1 PE_MIX_WG_PER_CU=1 -- --mfcc $prec 65536
done
run "one-tile-net-only-f64"    PE_PAIR=0 PE_FUSED_SKIP=1 -- --mfcc f64 65536
run "one-tile-frames-only-f64" PE_PAIR=0 PE_FUSED_SKIP=2 -- --mfcc f64 65536
