"""How much back-to-back work brings the GPU out of idle: the driver-style region (5 + 20 steps) after 1 s of idle
followed by n launches (no host synchronisation in between).   python tools/gpu_cold_start2.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
K, W, B = 20, 5, 4096
dev = torch.device('cuda', 0)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B)
pcm = (torch.randn((64, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros((K, B), device=dev)
st = torch.cuda.current_stream().cuda_stream
base, ob = pcm.data_ptr(), out.data_ptr()

def launches(n, o=0):
    for i in range(n):
        eng.update_device(base + (i % 64) * B * 2048, 1024, ob + (i % K) * B * 4 * o, st)

def region():
    launches(W)
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    launches(K, 1)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / K

for pre in (0, 100, 300, 1000, 3000, 10000, 30000):
    r = []
    for rep in range(3):
        time.sleep(0.5)
        t0 = time.perf_counter()
        launches(pre)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r.append(region())
    print('%6d launches (%.1f ms) before the region: %s us/step' % (pre, 1e3 * dt, ' '.join('%.2f' % x for x in r)))
eng.close()
