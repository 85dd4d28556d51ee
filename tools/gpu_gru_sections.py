"""Shader-clock stamps of the critical wave of gru_tile_cw INSIDE the product's launches (variant library `grut` =
-DPE_GRU_TIMERS -DPE_TUNING): prologue and per-timestep cost of the network role, fused with the MFCC roles or alone
(PE_FUSED_SKIP=1).    python tools/gpu_gru_sections.py [streams]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import _lib, synth
from mycroft_precise_amd.params import pr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dbg = os.path.join(REPO, 'mycroft_precise_amd', 'csrc', 'build', 'variants', 'libprecise_engine_grut.so')
_lib._lib = None
_lib.LIB_PATH = dbg
lib = _lib.load()
raw = ctypes.CDLL(dbg)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B)
dev = torch.device('cuda', 0)
pcm = (torch.randn((64, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros(B, device=dev)
stream = torch.cuda.current_stream().cuda_stream
rows = []
for u in range(64):
    eng.update_device(pcm[u].data_ptr(), 1024, out.data_ptr(), stream)
    torch.cuda.synchronize()
    t = (ctypes.c_ulonglong * (256 * 32))()
    raw.pe_debug_read_gru_timers(t, 256 * 32)
    t = np.array(t[:], dtype=np.int64).reshape(256, 32)
    if u >= 32:
        q, kc, ke = eng.stream_state()
        rows.append((int(kc[0]), np.mean(t[:, 1] - t[:, 0]), np.mean(t[:, 8] - t[:, 1]) / 29.0, np.mean(t[:, 9] - t[:, 0]),
                     np.mean(t[:, 3] - t[:, 2]), np.mean(t[:, 4] - t[:, 3]), np.mean(t[:, 5] - t[:, 4]), np.mean(t[:, 6] - t[:, 5])))
rows = np.array(rows)
two = np.diff(np.concatenate([[rows[0, 0] - 1], rows[:, 0]])) >= 2
for name, sel in (('one-frame updates', ~two), ('two-frame updates', two)):
    r = rows[sel]
    if len(r):
        print('%-18s (%2d): prologue %5.0f cycles, per timestep %5.0f, kernel top -> result %6.0f;  step 10: phase 1 %4.0f, phase 2 %4.0f, barrier %4.0f, blend + publish %4.0f'
              % (name, len(r), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), r[:, 4].mean(), r[:, 5].mean(), r[:, 6].mean(), r[:, 7].mean()))
eng.close()
