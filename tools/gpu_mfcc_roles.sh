for sk in 0 1 2; do echo "== PE_MFCC_SKIP=$sk"; PE_MFCC_SKIP=$sk python tools/gpu_sizes.py 4096 65536 2>&1 | grep streams | sed 's/network alone.*//' ; done
