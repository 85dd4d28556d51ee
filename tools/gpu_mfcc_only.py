"""MFCC stage N times at one size, for rocprofv3 runs.
    python tools/gpu_mfcc_only.py <streams> [n] [f64|f32] [vectors|update|keep]
vectors (default): pe_update_vectors_device, the MFCC launch alone; update / keep: whole updates through pe_update_device /
pe_update_device_keep (above 16 384 streams: mfcc_kernel + the network kernel, the profiler separates them by name)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
B = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
prec = sys.argv[3] if len(sys.argv) > 3 else 'f64'
how = sys.argv[4] if len(sys.argv) > 4 else 'vectors'
dev = torch.device('cuda', 0)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B, mfcc_precision=prec)
pcm = (torch.randn((16, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
for i in range(n):
    if how == 'vectors':
        eng.update_vectors_device(pcm[i % 16].data_ptr(), 1024, 0, st)
    else:
        eng.update_device(pcm[i % 16].data_ptr(), 1024, out.data_ptr(), st, keep=(how == 'keep'))
torch.cuda.synchronize()
eng.close()
