"""MFCC section timers of ONE wave (wave 0 of workgroup 0; debug library from tools/build_debug.sh): shader-clock
stamps along mfcc_frame_tasks / mfcc_wave_frame for a few steady-state updates.
    python tools/gpu_sections.py [streams]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from mycroft_precise_amd import _lib, synth
from mycroft_precise_amd.params import pr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dbg = os.environ.get('PE_DBG_LIB') or os.path.join(REPO, 'mycroft_precise_amd', 'csrc', 'build', 'libprecise_engine_dbg.so')
_lib._lib = None
_lib.LIB_PATH = dbg
lib = _lib.load()
raw = ctypes.CDLL(dbg)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B)
eng.set_fused(False)
pcm = np.random.default_rng(0).integers(-3000, 3000, (16, B, 1024)).astype('<i2')
names = {0: 'kernel top', 1: 'tables in LDS', 2: 'first PCM converted', 3: 'frame start', 4: 'fft done', 5: 'mirror done',
         6: 'power in LDS', 7: 'mel partials', 8: 'log done', 9: 'dct done', 10: 'row stored', 15: 'wave end'}
for u in range(16):
    eng.update_vectors(pcm[u], want_features=False)
    t = (ctypes.c_ulonglong * 32)()
    raw.pe_debug_read_timers(t, 32)
    t = np.array(t[:], dtype=np.int64)
    if u >= 10:
        q, kc, ke = eng.stream_state()
        order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15]
        print('update %d (frames so far %d): ' % (u, kc[0]) + ', '.join('%s +%d' % (names[k], t[k] - t[0]) for k in order[1:]))
eng.close()
