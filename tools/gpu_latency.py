"""Per-call latency of the host entry points at small batch (BASELINE configs[0]: one stream through Listener.update)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from mycroft_precise_amd import synth
from mycroft_precise_amd.network_runner import Listener, BatchedListener
from mycroft_precise_amd.model import save_weights

w = synth.make_weights()
path = '/tmp/lat_model.npz'
save_weights(path, w)
lis = Listener(path, 2048)
data = synth.stream_pcm(0, 1024 * 400).tobytes()
for i in range(50):
    lis.update(data[i * 2048:(i + 1) * 2048])
t0 = time.perf_counter()
n = 300
for i in range(50, 50 + n):
    lis.update(data[i * 2048:(i + 1) * 2048])
dt = (time.perf_counter() - t0) / n
print('Listener.update, 1 stream, 2048-byte chunks: %.1f us per call (real-time budget 64000 us)' % (dt * 1e6))
for B in (1, 16, 256, 4096):
    bl = BatchedListener(w, B)
    pcm = np.stack([synth.stream_pcm(s % 50, 1024 * 40).reshape(40, 1024) for s in range(min(B, 50))], axis=1)
    pcm = np.ascontiguousarray(np.tile(pcm, (1, (B + 49) // 50, 1))[:, :B])
    for u in range(10):
        bl.update(pcm[u])
    t0 = time.perf_counter()
    for u in range(10, 40):
        bl.update(pcm[u])
    dt = (time.perf_counter() - t0) / 30
    print('BatchedListener.update (host buffers, PCIe both ways), %5d streams: %.1f us per call = %.2f M windows/s' % (B, dt * 1e6, B / dt / 1e6))
