"""Do the batched MFCC launch and the batched network launch overlap when issued on two HIP streams?  Two engines
(independent state) run pe_update_many on two streams; compare with the same work on one stream."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr

B, depth = 4096, 8
dev = torch.device('cuda', 0)
w = synth.make_weights()
engs = [_lib.HipEngine(pr, w, n_streams=B) for _ in range(2)]
for e in engs:
    e.reserve_updates(depth, 1024)
pcm = (torch.randn((32, B, 1024), device=dev) * 3000).to(torch.int16)
outs = [torch.zeros((depth, B), device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(two_streams, rounds=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(rounds):
        for k, e in enumerate(engs):
            st = streams[k if two_streams else 0]
            e.update_many_device(pcm[(i * depth) % 24].data_ptr(), 1024, depth, outs[k].data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (rounds * 2 * depth) * 1e6


for two in (False, True, False, True):
    us = run(two)
    print('%s: %.2f us per update of %d streams = %.1f M windows/s' % ('two streams' if two else 'one stream ', us, B, B / us))
