#!/usr/bin/env python3
"""GPU-box diagnostic: stage-by-stage comparison of the HIP path with the oracle (checker only).
Run through gpurun; prints max errors per stage and writes gpurun_out/diag.json."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from mycroft_precise_amd import synth                      # noqa: E402
from mycroft_precise_amd._lib import HipEngine             # noqa: E402
from mycroft_precise_amd.params import pr                   # noqa: E402
from oracle import listener as ol, keras_gru, sonopy_restated as so   # noqa: E402

out = {}
w = synth.make_weights()


def report(name, a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    out[name] = {'max_abs': float(d.max()) if d.size else 0.0, 'shape': list(a.shape)}
    print('%-40s max|d|=%.3e  (ref max %.3e) shape=%s' % (name, out[name]['max_abs'], np.abs(b).max() if b.size else 0, a.shape), flush=True)
    return d


for prec in ('f64', 'f32'):
    # 1. stateless MFCC
    eng = HipEngine(pr, w, n_streams=5, mfcc_precision=prec)
    audio = synth.stream_pcm(3, 16000).astype(np.float64) / 32768.0
    got = eng.vectorize_raw(audio)
    ref = so.mfcc_spec(audio, 16000, (1600, 800), num_filt=20, fft_size=512, num_coeffs=13)
    d = report(prec + ' vectorize_raw', got, ref)
    if d.max() > 1e-3:
        print('got[0]', got[0]); print('ref[0]', ref[0])

    # 2. GRU predict on explicit features
    rng = np.random.default_rng(1)
    feats = rng.normal(0, 2, (37, 29, 13)).astype(np.float32)
    report(prec + ' predict', eng.predict(feats)[:, 0], keras_gru.predict(feats, w)[:, 0])

    # 3. streaming: features + raw outputs over 40 updates, 5 streams incl. zeros / square
    kinds = ['tone_noise', 'tone_noise', 'zeros', 'square', 'quiet']
    n_up = 40
    pcm = np.stack([synth.stream_pcm(s, n_up * 1024, k).reshape(n_up, 1024) for s, k in enumerate(kinds)], axis=1)
    bo = ol.BatchedOracle(w, 5)
    worst_f, worst_r = 0.0, 0.0
    for u in range(n_up):
        raw = eng.update(pcm[u])
        feats_dev = eng.get_vectors()
        raw_ref = bo.update_raw(pcm[u])
        df = np.abs(feats_dev.astype(np.float64) - bo.mfccs).max()
        dr = np.abs(raw.astype(np.float64) - raw_ref).max()
        if u < 4 or df > 1e-3 or dr > 1e-4:
            print('  update %2d  max|dfeat|=%.3e  max|draw|=%.3e  raw=%s' % (u, df, dr, np.round(raw, 5)), flush=True)
        worst_f, worst_r = max(worst_f, df), max(worst_r, dr)
    out[prec + ' stream'] = {'feat': worst_f, 'raw': worst_r}
    print('%s streaming: worst feature diff %.3e, worst raw diff %.3e' % (prec, worst_f, worst_r), flush=True)
    q, kc, ke = eng.stream_state()
    print('  state q=%s kc=%s ke=%s' % (q, kc, ke))
    eng.close()

# 4. odd chunk sizes against the single-stream oracle
for cb in (500, 1600, 3000, 10000, 48000):
    eng = HipEngine(pr, w, n_streams=1)
    lis = ol.OracleListener(w)
    pcm = synth.stream_pcm(7, 48000)
    worst = 0.0
    for off in range(0, len(pcm) - cb + 1, cb):
        raw = float(eng.update(pcm[off:off + cb].reshape(1, -1))[0])
        ref = lis.update_raw(pcm[off:off + cb].tobytes())
        worst = max(worst, abs(raw - ref))
    out['chunk_%d' % cb] = worst
    print('chunk %5d samples: worst raw diff %.3e' % (cb, worst), flush=True)
    eng.close()

# 5. quick timing at B=4096
B = 4096
eng = HipEngine(pr, w, n_streams=B)
pcm = np.random.default_rng(0).integers(-3000, 3000, (B, 1024)).astype('<i2')
for _ in range(3):
    eng.update(pcm)
eng.set_timing(True)
ts = []
for _ in range(10):
    eng.update(pcm)
    ts.append(eng.last_timing())
ts = np.array(ts)
print('B=4096 kernel times (ms): mfcc median %.4f  gru median %.4f' % (np.median(ts[:, 0]), np.median(ts[:, 1])))
out['timing_ms_B4096'] = {'mfcc': float(np.median(ts[:, 0])), 'gru': float(np.median(ts[:, 1]))}
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
with open(os.path.join(REPO, 'gpurun_out', 'diag.json'), 'w') as f:
    json.dump(out, f, indent=1, default=float)
