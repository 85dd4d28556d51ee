import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
w = synth.make_weights()
for B, ring, n_up in ((8192, 'f32', 300), (8192, 'bf16', 300), (65536, 'bf16', 120), (4096, None, 300)):
    base = synth.batch_pcm(1, n_up)
    kw = dict(mfcc_precision='f32', gru_precision='bf16', ring_precision=ring) if ring else dict(mfcc_precision='f32')
    eng = _lib.HipEngine(pr, w, n_streams=B, **kw)
    bad = 0
    for u in range(n_up):
        raw = eng.update(np.repeat(base[u], B, axis=0))
        bad += int((raw != raw[0]).sum())
    feats = eng.get_vectors()
    fbad = int(np.any(feats != feats[0], axis=(1, 2)).sum())
    print('product library, float32 frames,', ('bf16 network, %s rows' % ring) if ring else 'float32 network', '%d streams x %d fused updates (%.1f M frames): positions that ever disagreed %d, feature windows that differ at the end %d'
          % (B, n_up, B * n_up * 1.28 / 1e6, bad, fbad), flush=True)
    eng.close()
