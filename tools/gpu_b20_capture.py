"""(round 4, bisecting aid; needs tools/build_variants.sh cap "-DPE_DBG_CAPTURE=1" and PE_LIB / PE_B20=1): one fused update with
the same audio on every stream; what each frame task loaded (4 dwords per lane) and produced, compared across streams."""
import os, sys, ctypes
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
w = synth.make_weights()
B = 8192
base = synth.batch_pcm(1, 3)
eng = _lib.HipEngine(pr, w, n_streams=B, mfcc_precision='f32', gru_precision='bf16', ring_precision='f32')
lib = ctypes.CDLL(os.environ['PE_LIB'])
lib.pe_dbg_capture_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
names = ['pcm a0', 'pcm a1', 'pcm a2', 'pcm a3', 'coefficient (lane 4c..4c+3)', 'log-mel LM[lane]', 'power P[lane]', 'partial PART[lane]',
         'regs after convert', 'after pass a', 'after exchange b (permlane swaps)', 'after pass b', 'after exchange c (LDS)', 'after pass c', 'after exchange d (LDS)', 'after pass d']
for u in range(3):
    eng.update(np.repeat(base[u], B, axis=0))
    cap = np.zeros((B, 2, 64, 16), np.uint32)
    rc = lib.pe_dbg_capture_read(cap.ctypes.data_as(ctypes.c_void_p), cap.nbytes)
    feats = eng.get_vectors()
    fbad = np.nonzero(np.any(feats != feats[0], axis=(1, 2)))[0]
    print('update', u, 'rc', rc, 'streams with wrong windows', len(fbad), fbad[:8])
    for par in (0, 1):
        ref = cap[0, par]
        diff = np.any(cap[:, par] != ref[None], axis=1)          # [B, 8]: which captured words differ anywhere over the lanes
        for k in range(16):
            rows = np.nonzero(diff[:, k])[0]
            if len(rows):
                s = rows[0]
                lanes = np.nonzero(cap[s, par, :, k] != ref[:, k])[0]
                print('  row parity %d  %-28s differs in %4d streams (first %d, lanes %s)' % (par, names[k], len(rows), s, lanes[:10]))
                if 4 <= k < 8:
                    print('      got ', cap[s, par, lanes[:4], k].view(np.float32), ' want', ref[lanes[:4], k].view(np.float32))
                else:
                    print('      got ', [hex(x) for x in cap[s, par, lanes[:4], k]], ' want', [hex(x) for x in ref[lanes[:4], k]])
eng.close()
