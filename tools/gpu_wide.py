"""Network-launch time of the wide / stacked configuration (pe_run_device on a filled ring), optionally on
an experiment build (PE_LIB=...).  usage: gpu_wide.py [units,units] [streams]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import _lib, synth
from mycroft_precise_amd.params import pr

if os.environ.get('PE_LIB'):
    _lib.LIB_PATH = os.environ['PE_LIB']
units = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (256, 256)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device('cuda:0')
w = synth.make_weights(pr.n_mfcc, units)
eng = _lib.HipEngine(pr, w, n_streams=B)
if os.environ.get('PE_WIDE_TILING'):
    eng.set_gru_tiling(int(os.environ['PE_WIDE_TILING']))      # 0 = f32-input MFMAs, 2 = float32 products on the bf16 pipe
print('form', eng.gru_tiling(), end='  ')
pcm = (torch.randn((40, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
for i in range(40):
    eng.update_vectors_device(pcm[i].data_ptr(), 1024, 0, st)
for i in range(3):
    eng.run_device(out.data_ptr(), st)
torch.cuda.synchronize()
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    eng.run_device(out.data_ptr(), st)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
H = units[0]
flop = 0
f_in = pr.n_mfcc
for h in units:
    flop += 2 * 3 * h * (f_in + h)
    f_in = h
# MFMA-issued flops (K padded to 16 for the layer-0 input)
fm = 2 * 3 * H * (16 + H) + (2 * 3 * H * (H + H) if len(units) == 2 else 0)
print('%s units=%s streams=%d: %.3f ms per network launch, %.1f TFLOP/s issued, checksum %.6f' % (
    os.path.basename(_lib.LIB_PATH), units, B, ms, fm * pr.n_features * B / ms / 1e9, float(out.sum())))
