"""(round 5) section timers of ONE wave of the four-frames-per-wave MFCC kernel (wave 0 of workgroup 0, its LAST pass):
shader-clock stamps along mfcc_quad_frames.  Library: tools/build_debug.sh quad "-DPE_TUNING -DPE_QUAD_WAVES=4".
    PE_QUAD=1 PE_QUAD_WG_PER_CU=2 python tools/gpu_quad_sections.py [streams] [f64|f32]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from mycroft_precise_amd import _lib, synth
from mycroft_precise_amd.params import pr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prec = sys.argv[2] if len(sys.argv) > 2 else 'f64'
dbg = os.environ.get('PE_DBG_LIB') or os.path.join(REPO, 'mycroft_precise_amd', 'csrc', 'build', 'libprecise_engine_dbg_quad.so')
_lib._lib = None
_lib.LIB_PATH = dbg
lib = _lib.load()
raw = ctypes.CDLL(dbg)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B, mfcc_precision=prec)
eng.set_fused(False)
pcm = np.random.default_rng(0).integers(-3000, 3000, (4, B, 1024)).astype('<i2')
names = {1: 'tables in LDS', 2: 'next pass fetched', 3: 'PCM converted', 4: 'first 16-point pass + twiddles', 5: 'transposed', 6: 'second 16-point pass',
         7: 'split, power in LDS', 8: 'mel partials', 9: 'log done', 10: 'dct done', 11: 'row stored', 15: 'wave end'}
for u in range(12):
    eng.update_vectors(pcm[u % 4], want_features=False)
    t = (ctypes.c_ulonglong * 32)()
    raw.pe_debug_read_timers(t, 32)
    t = np.array(t[:], dtype=np.int64)
    if u >= 8:
        order = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 15]
        print('update %d: ' % u + ', '.join('%s +%d' % (names[k], t[k] - t[0]) for k in order))
        print('     last pass, section by section: ' + ', '.join('%s %d' % (names[b], t[b] - t[a]) for a, b in zip([2, 3, 4, 5, 6, 7, 8, 9, 10], [3, 4, 5, 6, 7, 8, 9, 10, 11])))
eng.close()
