"""pe_update_many throughput matrix (streams x depth x precision), per-kernel split via HIP events off."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import _lib
from mycroft_precise_amd.params import pr
from mycroft_precise_amd import synth

if os.environ.get('PE_LIB'):
    _lib.LIB_PATH = os.environ['PE_LIB']
dev = torch.device('cuda:0')
w = synth.make_weights()


def bench_many(B, depth, prec='f64', gru='f32', rounds=40):
    eng = _lib.HipEngine(pr, w, n_streams=B, mfcc_precision=prec, gru_precision=gru)
    eng.reserve_updates(depth, 1024)
    n_res = max(64, 2 * depth)
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


BS = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [4096, 16384]
DEPTHS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2, 4, 8, 16, 32]
for B in BS:
    for depth in DEPTHS:
        for prec, gru in (('f64', 'f32'), ('f32', 'bf16')):
            wall = bench_many(B, depth, prec=prec, gru=gru, rounds=40 if depth <= 8 else 20)
            print('%-8d MANY depth=%-3d mfcc=%s gru=%-5s | %8.2f us/update | %10.1f Mwin/s' % (B, depth, prec, gru, wall, B / wall), flush=True)
