#!/bin/bash
# fused / MFCC-alone time against the number of resident frame workgroups per compute unit (PE_FRAME_WG_PER_CU)
for n in 1 2 3 4; do echo "== PE_FRAME_WG_PER_CU=$n"; PE_FRAME_WG_PER_CU=$n python tools/gpu_sizes.py "$@" 2>&1 | grep streams | sed 's/network alone.*//'; done
