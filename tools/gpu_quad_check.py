"""(round 5) four-frames-per-wave MFCC (PE_QUAD=1, tuning build): features against the oracle + time of the MFCC launch.
    PE_QUAD=1 PE_LIB=.../libprecise_engine_quad.so python tools/gpu_quad_check.py [streams] [f64|f32]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
from oracle import listener as ol
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prec = sys.argv[2] if len(sys.argv) > 2 else 'f64'
w = synth.make_weights()
# parity: 37 streams, 40 updates, two launches (so that the MFCC launch is the quad kernel when PE_QUAD=1)
n, n_up = 37, 40
pcm = synth.batch_pcm(n, n_up)
eng = _lib.HipEngine(pr, w, n_streams=n, mfcc_precision=prec)
eng.set_fused(False)
ref = ol.BatchedOracle(w, n)
worst_f, worst_p = 0.0, 0.0
for u in range(n_up):
    raw = eng.update(pcm[u])
    want = ref.update_raw(pcm[u])
    worst_p = max(worst_p, float(np.abs(raw.astype(np.float64) - want).max()))
    worst_f = max(worst_f, float(np.abs(eng.get_vectors().astype(np.float64) - ref.mfccs).max()))
print('PE_QUAD=%s %s: max |feature - oracle| %.3g, max |p - oracle| %.3g' % (os.environ.get('PE_QUAD', '0'), prec, worst_f, worst_p))
eng.close()
dev = torch.device('cuda', 0)
eng = _lib.HipEngine(pr, w, n_streams=B, mfcc_precision=prec)
eng.set_fused(False)
x = (torch.randn((16, B, 1024), device=dev) * 3000).to(torch.int16)
st = torch.cuda.current_stream().cuda_stream
for i in range(20):
    eng.update_vectors_device(x[i % 16].data_ptr(), 1024, 0, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 200
for i in range(N):
    eng.update_vectors_device(x[i % 16].data_ptr(), 1024, 0, st)
torch.cuda.synchronize()
print('   MFCC launch alone at %d streams: %.2f us' % (B, (time.perf_counter() - t0) / N * 1e6))
eng.close()
