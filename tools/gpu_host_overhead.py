"""Where the fixed cost of a SHORT timed region goes (the driver times 20 steps = 0.35 ms): host stamps around the same
sequence bench.py runs -- synchronize, K launches, wait -- for several waiting styles.
    python tools/gpu_host_overhead.py [K]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 4096
dev = torch.device('cuda', 0)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B)
pcm = (torch.randn((64, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros((K, B), device=dev)
st = torch.cuda.current_stream().cuda_stream
base, ob = pcm.data_ptr(), out.data_ptr()
for i in range(40):
    eng.update_device(base + (i % 64) * B * 2048, 1024, ob, st)
torch.cuda.synchronize()
for style in ('sync', 'spin', 'spin', 'sync', 'spin_prerec'):
    rows = []
    for rep in range(6):
        for i in range(5):
            eng.update_device(base + (i % 64) * B * 2048, 1024, ob, st)
        torch.cuda.synchronize()
        ev = torch.cuda.Event()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        eng.update_device(base, 1024, ob, st)
        ta = time.perf_counter()
        for i in range(1, K):
            eng.update_device(base + (i % 64) * B * 2048, 1024, ob + i * B * 4, st)
        tb = time.perf_counter()
        e1.record()
        if style.startswith('spin'):
            while not e1.query():
                pass
        tc = time.perf_counter()
        torch.cuda.synchronize()
        td = time.perf_counter()
        rows.append((1e6 * (ta - t0), 1e6 * (tb - t0), 1e6 * (tc - t0), 1e6 * (td - t0), 1e3 * e0.elapsed_time(e1)))
    r = np.median(np.array(rows), axis=0)
    print('%-12s first launch returned %6.1f us | all %d queued %6.1f | wait done %6.1f | synchronize done %6.1f | GPU e0->e1 %6.1f us  => %.2f us/step host, %.2f GPU'
          % (style, r[0], K, r[1], r[2], r[3], r[4], r[3] / K, r[4] / K))
eng.close()
