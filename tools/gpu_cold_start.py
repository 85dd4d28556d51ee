"""How much of a SHORT timed region (the driver's --warmup 5 --steps 20) is the GPU coming out of idle: the same
region timed after an idle second, right after another region, and after half a second of launches; per-step GPU
durations (HIP events) of the cold case.   python tools/gpu_cold_start.py [spin]
`spin`: hipSetDeviceFlags(hipDeviceScheduleSpin) before the first HIP call of the process."""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
if len(sys.argv) > 1 and sys.argv[1] == 'spin':
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))      # the runtime torch loaded
    print('hipSetDeviceFlags(spin) ->', hip.hipSetDeviceFlags(1))
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
K, W, B = 20, 5, 4096
dev = torch.device('cuda', 0)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B)
pcm = (torch.randn((64, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros((K, B), device=dev)
st = torch.cuda.current_stream().cuda_stream
base, ob = pcm.data_ptr(), out.data_ptr()

def region(n_warm=W, events=False):
    for i in range(n_warm):
        eng.update_device(base + (i % 64) * B * 2048, 1024, ob, st)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)] if events else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        if events: evs[i].record()
        eng.update_device(base + (i % 64) * B * 2048, 1024, ob + i * B * 4, st)
    if events: evs[K].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [1e3 * evs[i].elapsed_time(evs[i + 1]) for i in range(K)] if events else None
    return 1e6 * dt / K, per

def busy(seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for i in range(64):
            eng.update_device(base + i * B * 2048, 1024, ob, st)
        torch.cuda.synchronize()

for rep in range(3):
    time.sleep(1.0)
    cold, _ = region()
    warm, _ = region()
    busy(0.5)
    hot, _ = region()
    gaps = []
    for gap in (0.001, 0.01, 0.1):
        busy(0.3)
        time.sleep(gap)
        gaps.append(region()[0])
    print('us/step  after 1 s idle %.2f | right after a region %.2f | after 0.5 s of launches %.2f | 0.3 s of launches then idle 1 / 10 / 100 ms: %.2f %.2f %.2f'
          % ((cold, warm, hot) + tuple(gaps)))
time.sleep(1.0)
cold, per = region(events=True)
print('cold with events %.2f us/step; per-step GPU us:' % cold, ' '.join('%.1f' % p for p in per))
hot, per = region(events=True)
print('warm with events %.2f us/step; per-step GPU us:' % hot, ' '.join('%.1f' % p for p in per))
eng.close()
