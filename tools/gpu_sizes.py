"""Per-size timing of the three pieces on one MI355X: fused update, MFCC alone (pe_update_vectors_device), network
alone (pe_run_device), wall clock over back-to-back launches.
    python tools/gpu_sizes.py [--mfcc f64|f32] [--gru f32|bf16] [sizes...]"""
import os, sys, time, argparse
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr

ap = argparse.ArgumentParser()
ap.add_argument('--mfcc', default='f64')
ap.add_argument('--gru', default='f32')
ap.add_argument('--ring', default='f32')
ap.add_argument('--desync', action='store_true')
ap.add_argument('--proj', type=int, default=-1)
ap.add_argument('--waves', type=int, default=0)
ap.add_argument('sizes', nargs='*', type=int, default=[4096, 8192, 16384, 65536])
args = ap.parse_args()
dev = torch.device('cuda', 0)
w = synth.make_weights()
rng = np.random.default_rng(3)
for B in args.sizes:
    eng = _lib.HipEngine(pr, w, n_streams=B, mfcc_precision=args.mfcc, gru_precision=args.gru, ring_precision=args.ring)
    if args.proj >= 0:
        eng.set_input_projection(bool(args.proj))
    if args.waves:
        eng.set_gru_waves(args.waves)
    n_res = 32
    pcm = (torch.randn((n_res, B, 1024), device=dev) * 3000).to(torch.int16)
    out = torch.zeros(B, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(40):
        eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st)
        if args.desync and i < 25:
            torch.cuda.synchronize()
            eng.clear(rng.random(B) < 0.2)
    torch.cuda.synchronize()

    def timeit(fn, n=200):
        for i in range(10):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    fused = timeit(lambda i: eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st))
    mfcc = timeit(lambda i: eng.update_vectors_device(pcm[i % n_res].data_ptr(), 1024, 0, st))
    gru = timeit(lambda i: eng.run_device(out.data_ptr(), st))
    print('streams %6d  mfcc=%s gru=%s ring=%s%s proj=%d waves=%d: fused %7.2f us (%6.1f M windows/s)  mfcc alone %7.2f us (%6.1f M/s, %5.1f %% of 8 TB/s)  '
          'network alone %7.2f us (%6.1f M/s)' % (B, args.mfcc, args.gru, args.ring, ' desync' if args.desync else '', args.proj, args.waves, fused, B / fused, mfcc, B / mfcc,
                                                  100 * 2114.6 * B / (mfcc * 1e-6) / 8e12, gru, B / gru), flush=True)
    eng.close()
