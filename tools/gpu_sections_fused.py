"""MFCC section timers of ONE frame wave inside the FUSED launch without the network role (PE_FUSED_SKIP=2, variant library
`dbgt` = -DPE_SECTION_TIMERS -DPE_TUNING): where the frame role's time goes at 4096 streams, two streams per wave.
    PE_FUSED_SKIP=2 python tools/gpu_sections_fused.py [streams]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import _lib, synth
from mycroft_precise_amd.params import pr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dbg = os.path.join(REPO, 'mycroft_precise_amd', 'csrc', 'build', 'variants', 'libprecise_engine_dbgt.so')
_lib._lib = None
_lib.LIB_PATH = dbg
lib = _lib.load()
raw = ctypes.CDLL(dbg)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B)
dev = torch.device('cuda', 0)
pcm = (torch.randn((24, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
names = {0: 'top', 1: 'tables+first pcm issued', 2: 'pcm converted', 3: 'frame start', 4: 'fft', 5: 'mirror', 6: 'power', 7: 'mel', 8: 'log', 9: 'dct', 10: 'row', 15: 'wave end'}
for u in range(24):
    eng.update_device(pcm[u].data_ptr(), 1024, out.data_ptr(), st)
    torch.cuda.synchronize()
    t = (ctypes.c_ulonglong * 32)()
    raw.pe_debug_read_timers(t, 32)
    t = np.array(t[:], dtype=np.int64)
    if u >= 16:
        nw = 2048
        wt = (ctypes.c_ulonglong * (4 * nw))()
        raw.pe_debug_read_wave_times(wt, nw)
        wt = np.array(wt[:], dtype=np.int64).reshape(nw, 4)
        ok = wt[:, 1] > wt[:, 0]
        ghz = (wt[ok, 3] - wt[ok, 2]) / ((wt[ok, 1] - wt[ok, 0]) * 10.0) / 1e0
        print('   shader clock over a frame wave\'s lifetime (cycle counter / wall clock): median %.2f GHz, min %.2f, max %.2f' % (np.median(ghz) / 1e3 * 1e3, ghz.min(), ghz.max()))
        st_, en = wt[ok, 0], wt[ok, 1]
        t00 = st_.min()
        print('   %d frame waves (100 MHz clock, us after the first wave start): starts min 0 / median %.2f / max %.2f; ends min %.2f / median %.2f / max %.2f; durations median %.2f / max %.2f'
              % (ok.sum(), np.median(st_ - t00) / 100, (st_.max() - t00) / 100, (en.min() - t00) / 100, np.median(en - t00) / 100, (en.max() - t00) / 100,
                 np.median(en - st_) / 100, (en - st_).max() / 100))
        simd = wt[ok, 2] & 3
        dur = (wt[ok, 1] - wt[ok, 0]) / 100.0
        print('   frame-wave duration by the SIMD it sits on (median / max us): ' + ', '.join('SIMD %d: %.2f / %.2f (%d waves)' % (k, np.median(dur[simd == k]), dur[simd == k].max(), (simd == k).sum()) for k in range(4)))
        q, kc, ke = eng.stream_state()
        order = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15]
        print('update %d (frames so far %d): ' % (u, kc[0]) + ', '.join('%s +%d' % (names[k], t[k] - t[0]) for k in order))
eng.close()
