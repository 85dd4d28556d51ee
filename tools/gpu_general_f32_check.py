import os, sys, warnings
sys.path.insert(0, '/root/repo')
import numpy as np
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd import params as P
from oracle import listener as ol
for kw in (dict(n_fft=400, n_filt=26, n_mfcc=13), dict(n_fft=256, n_filt=20, n_mfcc=13), dict(n_fft=1024, n_filt=40, n_mfcc=16), None):
    for chunk in (2889, 1024):
        for mfcc in ('f64', 'f32'):
            hpr, opr = P.pr, None
            w = synth.make_weights(seed=3)
            if kw:
                hpr = P.pr.copy(); hpr.__dict__.update(kw); opr = ol.Params(**kw)
                w = synth.make_weights(n_in=kw['n_mfcc'], units=(20,), seed=3)
            n, n_up = 72, 8
            kinds = ['tone_noise', 'tone_noise', 'quiet', 'square', 'zeros'] * 15
            pcm = np.stack([synth.stream_pcm(1000 + s, n_up * chunk, kinds[s]).reshape(n_up, chunk) for s in range(n)], axis=1)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                eng = _lib.HipEngine(hpr, w, n_streams=n, mfcc_precision=mfcc)
            ref = ol.BatchedOracle(w, n, opr) if opr else ol.BatchedOracle(w, n)
            wp, wf, where = 0.0, 0.0, None
            for u in range(n_up):
                got = eng.update(pcm[u]); want = ref.update_raw(pcm[u])
                d = np.abs(got - want)
                if d.max() > wp: wp = float(d.max()); where = (u, int(d.argmax()), kinds[int(d.argmax())])
                fd = np.abs(eng.get_vectors().astype(np.float64) - ref.mfccs)
                wf = max(wf, float(fd.max()))
            print(kw, chunk, mfcc, 'max |dp| %.3g at %s, max |dfeat| %.3g' % (wp, where, wf), flush=True)
            eng.close()
