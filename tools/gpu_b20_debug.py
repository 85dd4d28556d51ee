"""(round 4) which streams' FEATURE windows differ from stream 0 when every stream gets the same audio (fused bf16 launch)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
w = synth.make_weights()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n_up = 5
base = synth.batch_pcm(1, n_up)
eng = _lib.HipEngine(pr, w, n_streams=B, mfcc_precision=os.environ.get('DBG_MFCC', 'f32'), gru_precision='bf16', ring_precision='f32')
tot = 0
for u in range(n_up):
    raw = eng.update(np.repeat(base[u], B, axis=0))
    feats = eng.get_vectors()
    fbad = np.nonzero(np.any(feats != feats[0], axis=(1, 2)))[0]
    tot = len(fbad)
print(os.environ.get('PE_LIB', 'in-tree'), 'streams with wrong feature windows after %d updates: %d' % (n_up, tot), 'first', fbad[:6])
eng.close()
