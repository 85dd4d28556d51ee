"""Helper of test_gpu_parity.test_waves_owning_more_than_64_streams (run in a subprocess, because the resident
frame-workgroup count is read from the environment once per process): with PE_FRAME_WG_PER_CU=1 there are 1024 frame
waves, so B > 65536 streams gives every wave more than 64 streams -- the counter registers of mfcc_frame_tasks are
refilled batch by batch.  Size-independent checks: seeded streams against the oracle, identical input => identical
output wherever the copy sits (first or second batch of a wave), fused == two launches."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from mycroft_precise_amd import synth
from mycroft_precise_amd.network_runner import BatchedListener
from oracle import listener as ol

B, n_up, n_check = int(sys.argv[1]), 9, 12
w = synth.make_weights()
rng = np.random.default_rng(B)
base = synth.batch_pcm(n_check, n_up)
owner = rng.integers(0, n_check, B)
owner[:n_check] = np.arange(n_check)
owner[-n_check:] = np.arange(n_check)            # the tail of the batch: second counter batch of the last waves
hip = BatchedListener(w, B)
two = BatchedListener(w, B)
two.engine.set_fused(False)
ref = ol.BatchedOracle(w, n_check)
worst = 0.0
for u in range(n_up):
    pcm = base[u][owner]
    raw = hip.update_raw(pcm)
    want = ref.update_raw(base[u])
    assert raw.shape == (B,) and np.all(np.isfinite(raw))
    worst = max(worst, float(np.abs(raw[:n_check] - want).max()))
    assert np.array_equal(raw, raw[:n_check][owner]), u
    assert np.array_equal(raw, two.update_raw(pcm)), u
assert worst <= 2e-5, worst
for x, y in zip(hip.engine.stream_state(), two.engine.stream_state()):
    assert np.array_equal(x, y)
print('ok', B, worst)
