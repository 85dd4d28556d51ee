"""MFCC stage alone (pe_update_vectors_device) N times at one size, for rocprofv3 runs.
    python tools/gpu_mfcc_only.py <streams> [n] [f64|f32]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
B = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
prec = sys.argv[3] if len(sys.argv) > 3 else 'f64'
dev = torch.device('cuda', 0)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B, mfcc_precision=prec)
pcm = (torch.randn((16, B, 1024), device=dev) * 3000).to(torch.int16)
st = torch.cuda.current_stream().cuda_stream
for i in range(n):
    eng.update_vectors_device(pcm[i % 16].data_ptr(), 1024, 0, st)
torch.cuda.synchronize()
eng.close()
