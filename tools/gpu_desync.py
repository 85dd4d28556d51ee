"""Headline workload with desynchronised streams: streams restarted (masked pe_clear) at different updates have
different frame phases, so most tiles contain a stream that completes two frames in any given update."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr

B = 4096
dev = torch.device('cuda', 0)
w = synth.make_weights()
rng = np.random.default_rng(3)


def run(desync):
    eng = _lib.HipEngine(pr, w, n_streams=B)
    n_res = 64
    pcm = (torch.randn((n_res, B, 1024), device=dev) * 3000).to(torch.int16)
    out = torch.zeros(B, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(60):
        eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st)
        if desync and i < 25:
            torch.cuda.synchronize()
            eng.clear(rng.random(B) < 0.2)
    torch.cuda.synchronize()
    q, kc, ke = eng.stream_state()
    t0 = time.perf_counter()
    n = 200
    for i in range(n):
        eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    eng.close()
    return dt * 1e6, len(np.unique(q))


for desync in (False, True):
    us, phases = run(desync)
    print('desynchronised=%s: %d distinct leftover lengths, %.2f us per update, %.1f M windows/s' % (desync, phases, us, B / us))
