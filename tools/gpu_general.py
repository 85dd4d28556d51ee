"""General front end (non-stock ListenerParams) on one MI355X: per-update time of the MFCC launch alone, the network launch
alone and the whole update, for one parameter set.   python tools/gpu_general.py [--streams 4096] [--n-fft 1024 --n-filt 40 --n-mfcc 20]"""
import os, sys, time, argparse, warnings
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr

ap = argparse.ArgumentParser()
ap.add_argument('--streams', type=int, default=4096)
ap.add_argument('--n-fft', type=int, default=1024)
ap.add_argument('--n-filt', type=int, default=40)
ap.add_argument('--n-mfcc', type=int, default=20)
ap.add_argument('--mfcc', default='f64')
args = ap.parse_args()
hpr = pr.copy()
hpr.__dict__.update(dict(n_fft=args.n_fft, n_filt=args.n_filt, n_mfcc=args.n_mfcc))
w = synth.make_weights(n_in=args.n_mfcc)
B = args.streams
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    eng = _lib.HipEngine(hpr, w, n_streams=B, mfcc_precision=args.mfcc)
dev = torch.device('cuda', 0)
n_res = 32
pcm = (torch.randn((n_res, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
for i in range(40):
    eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st)
torch.cuda.synchronize()

def timeit(fn, n=100):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

upd = timeit(lambda i: eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st))
mfcc = timeit(lambda i: eng.update_vectors_device(pcm[i % n_res].data_ptr(), 1024, 0, st))
gru = timeit(lambda i: eng.run_device(out.data_ptr(), st))
print('%s streams %d n_fft=%d n_filt=%d n_mfcc=%d mfcc=%s: update %7.2f us (%6.1f M windows/s)  mfcc alone %7.2f us  network alone %7.2f us'
      % (os.environ.get('PE_LIB', 'product').split('_')[-1], B, args.n_fft, args.n_filt, args.n_mfcc, args.mfcc, upd, B / upd, mfcc, gru), flush=True)
eng.close()
