#!/bin/bash
# Round 4, second pass: does wave priority decide whether MFMA waves and VALU waves of one SIMD overlap?
#   tools/gpu_mix2.sh <tag>       (needs tools/build_variants.sh tun "")
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out; mkdir -p $OUT
tag=${1:-r4c}
echo "== pipe overlap"; timeout 200 tools/micro/build/pipe_overlap 2>&1 | tee $OUT/${tag}_pipe_overlap.csv
export PE_LIB=$PWD/mycroft_precise_amd/csrc/build/variants/libprecise_engine_tun.so
run() {  # label, env..., -- args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "-- $label: ${envs[*]} $*"
  env "${envs[@]}" timeout 300 python tools/gpu_sizes.py "$@" 2>&1 | grep streams | sed "s/^/[$label] /" | tee -a $OUT/${tag}_sizes.log
}
: > $OUT/${tag}_sizes.log
for prec in f64 f32; do
  for pn in 0 3; do for pf in 0 3; do
    run "mix-$prec-net$pn-frame$pf" PE_PAIR=1 PE_MIX_PRIO_NET=$pn PE_MIX_PRIO_FRAME=$pf -- --mfcc $prec 65536
  done; done
done
run "mix-f64-net0-frame3-32768" PE_PAIR=1 PE_MIX_PRIO_NET=0 PE_MIX_PRIO_FRAME=3 -- --mfcc f64 32768
