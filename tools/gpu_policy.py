"""Launch-policy matrix: fused x waves-per-tile x batch (and the bf16 network), wall time per update."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr

w = synth.make_weights()
dev = torch.device('cuda', 0)


def bench_cfg(B, fused, waves, prec='f64', gru='f32'):
    eng = _lib.HipEngine(pr, w, n_streams=B, mfcc_precision=prec, gru_precision=gru)
    eng.set_fused(fused)
    if waves:
        eng.set_gru_waves(waves)
    n_res = 48 if B <= 16384 else 16
    steps = 120 if B <= 16384 else 40
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
    eng.close()
    del pcm
    return wall


Bs = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [4096, 8192, 16384, 32768, 65536]
print('%-8s %-6s %-6s %-9s | %10s | %10s' % ('B', 'fused', 'waves', 'precision', 'us/update', 'Mwin/s'))
for B in Bs:
    for prec, gru in (('f64', 'f32'), ('f32', 'bf16')):
        for fused in (True, False):
            for waves in ((4, 1) if gru == 'f32' else (0,)):
                wall = bench_cfg(B, fused, waves, prec, gru)
                print('%-8d %-6s %-6s %-9s | %10.2f | %10.1f' % (B, fused, waves or '-', prec + '/' + gru, wall, B / wall), flush=True)
