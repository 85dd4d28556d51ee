"""MFCC stage alone (pe_update_vectors_device), timed by HIP events around a run of launches, for one or several builds of
the library (PE_LIB; symbols an older build lacks are skipped).    python tools/gpu_mfcc_ab.py <streams> [f64|f32] [n]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import ctypes
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
if os.environ.get('PE_LIB'):
    probe = ctypes.CDLL(os.environ['PE_LIB'])
    for name in list(_lib.EXPORTS):
        if not hasattr(probe, name):
            _lib.EXPORTS.pop(name)
B = int(sys.argv[1]); prec = sys.argv[2] if len(sys.argv) > 2 else 'f64'; n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device('cuda', 0)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B, mfcc_precision=prec)
pcm = (torch.randn((25, B, 1024), device=dev) * 3000).to(torch.int16)
st = torch.cuda.current_stream().cuda_stream
def run(k):
    for i in range(k):
        eng.update_vectors_device(pcm[i % 25].data_ptr(), 1024, 0, st)
run(50)
torch.cuda.synchronize()
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(n); e1.record(); e1.synchronize()
    res.append(1e3 * e0.elapsed_time(e1) / n)
print('%s %d streams %s: %s us per MFCC launch' % (os.path.basename(os.environ.get('PE_LIB', 'in-tree')), B, prec, ' '.join('%.2f' % r for r in res)))
eng.close()
