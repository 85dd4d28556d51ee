cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for v in unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  timeout 300 python bench.py --no-cpu-baseline --no-batched --no-extra-configs > gpurun_out/ka_$v.json 2> gpurun_out/ka_$v.err
  python3 - gpurun_out/ka_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('HIP_FORCE_DEV_KERNARG=%-6s %7.1f M/s  step %.2f us  fused %.2f  mfcc %.2f  gru %.2f' % (sys.argv[2], d['value'] / 1e6, d['ms_per_step'] * 1e3,
      d['roofline']['avg_launch_ms'] * 1e3, d['roofline_mfcc']['avg_launch_ms'] * 1e3, d['roofline_gru']['avg_launch_ms'] * 1e3))
PY
done; done
