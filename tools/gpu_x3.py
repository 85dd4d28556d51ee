"""(round 4) gru_precision='x3' -- the float32 network as three bf16 pieces per operand on the XDL pipe -- against the
float32-MFMA engine, the numpy oracle (float32) and a float64 evaluation of the same windows; then timings by size.
    python tools/gpu_x3.py [sizes...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
from oracle import keras_gru

dev = torch.device('cuda', 0)
w = synth.make_weights()
B = 512
n_up = 40
pcm = synth.batch_pcm(B, n_up)
ea = _lib.HipEngine(pr, w, n_streams=B, gru_precision='f32')
eb = _lib.HipEngine(pr, w, n_streams=B); eb.set_gru_tiling(2)
worst = {'x3_vs_f32': 0.0, 'x3_vs_oracle32': 0.0, 'f32_vs_oracle32': 0.0, 'x3_vs_f64': 0.0, 'f32_vs_f64': 0.0, 'oracle32_vs_f64': 0.0}
for u in range(n_up):
    pa = ea.update(pcm[u])
    pb = eb.update(pcm[u])
    feats = eb.get_vectors()                      # [B][T][F] float32 window after the update
    o32 = keras_gru.predict(feats, w)[:, 0].astype(np.float64)
    o64 = keras_gru.predict(feats, w, dtype=np.float64)[:, 0]
    pa = pa.astype(np.float64); pb = pb.astype(np.float64)
    for k, d in (('x3_vs_f32', pb - pa), ('x3_vs_oracle32', pb - o32), ('f32_vs_oracle32', pa - o32), ('x3_vs_f64', pb - o64), ('f32_vs_f64', pa - o64), ('oracle32_vs_f64', o32 - o64)):
        worst[k] = max(worst[k], float(np.abs(d).max()))
print('max |dp| over %d streams x %d updates:' % (B, n_up), {k: '%.3g' % v for k, v in worst.items()}, flush=True)
# predict (explicit batch) path
x = eb.get_vectors()
pp = eb.predict(x) if hasattr(eb, 'predict') else None
if pp is not None:
    print('predict vs ring: %.3g' % float(np.abs(np.asarray(pp, np.float64).ravel() - pb).max()), flush=True)
ea.close(); eb.close()

sizes = [int(s) for s in sys.argv[1:]] or [4096, 8192, 65536]
for Bn in sizes:
    for gru, fz in (('f32', 1), ('x3', 1), ('bf16', 1)):
        eng = _lib.HipEngine(pr, w, n_streams=Bn, gru_precision='f32' if gru == 'x3' else gru)
        if gru != 'bf16':
            eng.set_gru_tiling(2 if gru == 'x3' else 1 if Bn <= 8192 else 0)
        n_res = 16
        pcm_d = (torch.randn((n_res, Bn, 1024), device=dev) * 3000).to(torch.int16)
        out = torch.zeros(Bn, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for i in range(40):
            eng.update_device(pcm_d[i % n_res].data_ptr(), 1024, out.data_ptr(), st)
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
        upd = timeit(lambda i: eng.update_device(pcm_d[i % n_res].data_ptr(), 1024, out.data_ptr(), st))
        mf = timeit(lambda i: eng.update_vectors_device(pcm_d[i % n_res].data_ptr(), 1024, 0, st))
        net = timeit(lambda i: eng.run_device(out.data_ptr(), st))
        print('streams %6d gru=%-4s fused=%d: update %7.2f us (%6.1f M windows/s)  mfcc alone %7.2f us  network alone %7.2f us' % (Bn, gru, fz, upd, Bn / upd, mf, net), flush=True)
        eng.close()

# Do the network launch (XDL pipe) and the MFCC launch (vector pipe) of DIFFERENT kernels share the machine when issued
# on two HIP streams?  Two engines with independent state; per round one MFCC-only update and one network-only run.
for Bn in sizes:
    if Bn < 16384 or os.environ.get('X3_SKIP_STREAMS'):
        continue
    for gru, mfcc in (('x3', 'f64'), ('f32', 'f64'), ('bf16', 'f32')):
        ring = 'bf16' if gru == 'bf16' else 'f32'
        e_net = _lib.HipEngine(pr, w, n_streams=Bn, gru_precision='f32' if gru == 'x3' else gru, mfcc_precision=mfcc, ring_precision=ring)
        e_mf = _lib.HipEngine(pr, w, n_streams=Bn, gru_precision='f32' if gru == 'x3' else gru, mfcc_precision=mfcc, ring_precision=ring)
        if gru != 'bf16':
            e_net.set_gru_tiling(2 if gru == 'x3' else 0); e_mf.set_gru_tiling(2 if gru == 'x3' else 0)
        n_res = 16
        pcm_d = (torch.randn((n_res, Bn, 1024), device=dev) * 3000).to(torch.int16)
        out = torch.zeros(Bn, device=dev)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for i in range(40):
            e_net.update_device(pcm_d[i % n_res].data_ptr(), 1024, out.data_ptr(), s1.cuda_stream)
        torch.cuda.synchronize()

        def rounds(two, n=60):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                e_mf.update_vectors_device(pcm_d[i % n_res].data_ptr(), 1024, 0, (s2 if two else s1).cuda_stream)
                e_net.run_device(out.data_ptr(), s1.cuda_stream)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6
        r = [rounds(False), rounds(True), rounds(False), rounds(True)]
        print('streams %6d gru=%-4s mfcc=%s: MFCC + network per round, one stream %.1f / %.1f us, two streams %.1f / %.1f us' % (Bn, gru, mfcc, r[0], r[2], r[1], r[3]), flush=True)
        e_net.close(); e_mf.close()
