"""(round 5) one stream, one 2048-byte chunk per call: pe_update (pageable hipMemcpy, launch, hipMemcpy) against
pe_update_async + pe_wait on pinned and on pageable buffers -- C-ABI calls only, no Listener on top.
    python tools/gpu_single_latency.py [n_streams]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=n)
pcm = synth.batch_pcm(n, 64)
def bench(name, fn, reps=3000):
    for u in range(200): fn(u)
    t0 = time.perf_counter()
    for u in range(reps): fn(u)
    print('%-58s %7.2f us per call' % (name, (time.perf_counter() - t0) / reps * 1e6))
bench('pe_update (pageable in, pageable out)', lambda u: eng.update(pcm[u % 64]))
pin = eng.host_array((n, 1024), np.int16)
pout = eng.host_array((n,), np.float32)
def f_async_pinned(u):
    pin[...] = pcm[u % 64]
    eng.update_async(pin, pout); eng.wait()
bench('pe_update_async + pe_wait, pinned in / out (+ host copy)', f_async_pinned)
out = np.empty(n, np.float32)
def f_async_pageable(u):
    eng.update_async(pcm[u % 64], out); eng.wait()
bench('pe_update_async + pe_wait, pageable in / out', f_async_pageable)
eng.close()
