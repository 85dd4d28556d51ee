"""MFCC section timers of one workgroup (debug library from tools/build_debug.sh): cycles per section of
mfcc_stream_tile for the last four of twelve updates."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from mycroft_precise_amd import _lib, synth
from mycroft_precise_amd.params import pr

w = synth.make_weights()
# ---- MFCC section timers (debug library) ---------------------------------------------------------
dbg = os.path.join(REPO, 'mycroft_precise_amd', 'csrc', 'build', 'libprecise_engine_dbg.so')
if os.path.exists(dbg):
    _lib._lib = None
    _lib.LIB_PATH = dbg
    lib = _lib.load()
    raw = ctypes.CDLL(dbg)
    eng = _lib.HipEngine(pr, w, n_streams=4096)
    eng.set_fused(False)
    pcm = np.random.default_rng(0).integers(-3000, 3000, (12, 4096, 1024)).astype('<i2')
    names = ['start', 'tables->LDS', 'pcm loads', 'fft pass1+twiddle', 'transpose', 'fft pass2', 'mirror exchange',
             'power', 'mel+log', 'dct(+log)', 'frame loop end', 'carry+state']
    for u in range(12):
        eng.update(pcm[u])
        t = (ctypes.c_ulonglong * 32)()
        raw.pe_debug_read_timers(t, 32)
        t18 = int(t[18]); t = np.array(t[:12], dtype=np.int64)
        q, kc, ke = eng.stream_state()
        if u >= 8:
            d = np.diff(t)
            print('update %d (frames computed so far %d): total %d cycles' % (u, kc[0], t[11] - t[0]))
            print('   ' + ', '.join('%s=%d' % (n, v) for n, v in zip(names[1:], d)) + ', [mel sums only=%d]' % (t18 - t[7]))
    eng.close()
