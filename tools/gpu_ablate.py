#!/usr/bin/env python3
"""Time the stand-alone MFCC kernel (and the fused launch) of every ablation library; one process per
library (a process can bind only one)."""
import glob
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
import torch
from mycroft_precise_amd import synth, _lib
_lib.LIB_PATH = sys.argv[1]
from mycroft_precise_amd.params import pr
w = synth.make_weights(); B = 4096
dev = torch.device('cuda', 0)
pcm = (torch.randn((64, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros(B, device=dev); st = torch.cuda.current_stream().cuda_stream
res = []
for fused in (False, True):
    eng = _lib.HipEngine(pr, w, n_streams=B); eng.set_fused(fused)
    for i in range(40): eng.update_device(pcm[i %% 64].data_ptr(), 1024, out.data_ptr(), st)
    eng.set_timing(True); ev = []
    for i in range(100):
        eng.update_device(pcm[i %% 64].data_ptr(), 1024, out.data_ptr(), st); ev.append(eng.last_timing())
    ev = np.array(ev) * 1e3; res.append((np.median(ev[:, 0]), np.mean(ev[:, 0]), np.median(ev[:, 1]))); eng.close()
print('%%-12s mfcc median %%6.2f mean %%6.2f | gru %%6.2f | fused median %%6.2f mean %%6.2f' %% (os.path.basename(sys.argv[1])[10:-3], res[0][0], res[0][1], res[0][2], res[1][0], res[1][1]))
''' % REPO
for lib in sorted(glob.glob(os.path.join(REPO, 'mycroft_precise_amd', 'csrc', 'build', 'libpe_abl_*.so'))):
    subprocess.run([sys.executable, '-c', CHILD, lib], check=False)
