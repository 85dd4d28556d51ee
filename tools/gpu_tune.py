#!/usr/bin/env python3
"""GPU-box tuning harness: launch-shape matrix (fused x waves-per-tile x batch) with HIP-event and
wall timings, plus the MFCC section timers of the debug library.  Prints a table; not part of the product."""
import ctypes
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from mycroft_precise_amd import synth, _lib  # noqa: E402
from mycroft_precise_amd.params import pr  # noqa: E402

w = synth.make_weights()
dev = torch.device('cuda', 0)


def bench_cfg(B, fused, waves, prec='f64', steps=200, gru='f32'):
    eng = _lib.HipEngine(pr, w, n_streams=B, mfcc_precision=prec, gru_precision=gru)
    eng.set_fused(fused)
    eng.set_gru_waves(waves)
    n_res = 64
    pcm = (torch.randn((n_res, B, 1024), device=dev) * 3000).to(torch.int16)
    out = torch.zeros(B, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(40):
        eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e6
    eng.set_timing(True)
    ev = []
    for i in range(60):
        eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st)
        ev.append(eng.last_timing())
    ev = np.array(ev) * 1e3
    eng.close()
    return wall, np.median(ev[:, 0]), np.median(ev[:, 1])


print('%-8s %-6s %-6s %-5s | %10s %10s %10s | %12s' % ('B', 'fused', 'waves', 'prec', 'wall us', 'ev1 us', 'ev2 us', 'Mwin/s'))
for B in (4096,):
    for prec in ('f64',):
        for fused in (True, False):
            for waves in (4,):
                if B == 65536 and prec == 'f32' and not fused:
                    continue
                wall, e1, e2 = bench_cfg(B, fused, waves, prec)
                print('%-8d %-6s %-6d %-5s | %10.2f %10.2f %10.2f | %12.1f' % (B, fused, waves, prec, wall, e1, e2, B / wall), flush=True)

for B in (4096, 65536):
    for prec in ('f32', 'f64'):
        for fused in (True, False):
            wall, e1, e2 = bench_cfg(B, fused, 1, prec, gru='bf16')
            print('%-8d %-6s %-6s %-5s | %10.2f %10.2f %10.2f | %12.1f' % (B, fused, 'bf16', prec, wall, e1, e2, B / wall), flush=True)

def bench_many(B, depth, prec='f64', gru='f32', rounds=40):
    eng = _lib.HipEngine(pr, w, n_streams=B, mfcc_precision=prec, gru_precision=gru)
    eng.reserve_updates(depth, 1024)
    n_res = 64
    pcm = (torch.randn((n_res, B, 1024), device=dev) * 3000).to(torch.int16)
    out = torch.zeros((depth, B), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(6):
        eng.update_many_device(pcm[(i * depth) % (n_res - depth)].data_ptr(), 1024, depth, out.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(rounds):
        eng.update_many_device(pcm[(i * depth) % (n_res - depth)].data_ptr(), 1024, depth, out.data_ptr(), st)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / (rounds * depth) * 1e6
    eng.close()
    return wall


for B in (4096, 16384):
    for depth in (2, 4, 8, 16):
        for gru in ('f32', 'bf16'):
            wall = bench_many(B, depth, gru=gru)
            print('%-8d MANY depth=%-3d %-5s f64 | %10.2f us/update | %12.1f Mwin/s' % (B, depth, gru, wall, B / wall), flush=True)

# ---- MFCC section timers (debug library) ---------------------------------------------------------
dbg = os.path.join(REPO, 'mycroft_precise_amd', 'csrc', 'build', 'libprecise_engine_dbg.so')
if os.path.exists(dbg):
    _lib._lib = None
    _lib.LIB_PATH = dbg
    lib = _lib.load()
    raw = ctypes.CDLL(dbg)
    eng = _lib.HipEngine(pr, w, n_streams=4096)
    eng.set_fused(False)
    pcm = np.random.default_rng(0).integers(-3000, 3000, (12, 4096, 1024)).astype('<i2')
    names = ['start', 'tables->LDS', 'pcm loads', 'fft pass1+twiddle', 'transpose', 'fft pass2', 'mirror exchange',
             'power', 'mel+log', 'dct(+log)', 'frame loop end', 'carry+state']
    for u in range(12):
        eng.update(pcm[u])
        t = (ctypes.c_ulonglong * 32)()
        raw.pe_debug_read_timers(t, 32)
        t18 = int(t[18]); t = np.array(t[:12], dtype=np.int64)
        q, kc, ke = eng.stream_state()
        if u >= 8:
            d = np.diff(t)
            print('update %d (frames computed so far %d): total %d cycles' % (u, kc[0], t[11] - t[0]))
            print('   ' + ', '.join('%s=%d' % (n, v) for n, v in zip(names[1:], d)) + ', [mel sums only=%d]' % (t18 - t[7]))
    eng.close()
