"""(round 5; round 6: the pipeline keeps its leftovers in its own device chunks -- masked clears and subset updates added) randomized interleaving of the host-fed pipeline (pe_update_async / pe_wait, pinned and pageable buffers) with every
other entry point that has to drain it (pe_update, pe_update_many, pe_get_vectors, pe_predict, pe_clear), against an engine that only
ever takes synchronous updates: every probability and every feature window bit for bit.
    python tools/gpu_async_stress.py [seconds] [seed]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time()
rounds = ops = 0
while time.time() - t0 < budget:
    n = int(rng.choice([1, 7, 64, 300, 4096]))
    chunk = int(rng.choice([1024, 1024, 512, 2048, 800, 777]))
    kw = {}
    if rng.integers(0, 4) == 0:
        kw = dict(mfcc_precision='f32', gru_precision='bf16', ring_precision='bf16')
    w = synth.make_weights()
    a = _lib.HipEngine(pr, w, n_streams=n, **kw)          # synchronous updates only
    b = _lib.HipEngine(pr, w, n_streams=n, **kw)          # the pipeline
    depth = 3
    if rng.integers(0, 2):
        a.reserve_updates(depth, chunk); b.reserve_updates(depth, chunk)
        can_many = True
    else:
        can_many = False
    pins = [b.host_array((n, chunk), np.int16) for _ in range(4)]
    pouts = [b.host_array((n,), np.float32) for _ in range(4)]
    pending = []          # (out array of b, expected)
    def settle():
        global pending
        for out, want in pending:
            assert np.array_equal(out, want), ('async result differs', n, chunk, kw)
        pending = []
    base = synth.batch_pcm(min(n, 64), 8)                 # [8][<=64][1024]
    def next_pcm():
        u = int(rng.integers(0, 8))
        x = base[u][np.arange(n) % base.shape[1]]
        if chunk <= 1024:
            return np.ascontiguousarray(x[:, :chunk])
        return np.ascontiguousarray(np.concatenate([x, base[(u + 1) % 8][np.arange(n) % base.shape[1]]], axis=1)[:, :chunk])
    slot = 0
    for step in range(int(rng.integers(20, 60))):
        op = int(rng.integers(0, 10))
        ops += 1
        if op <= 4:                                       # asynchronous update, pinned or pageable
            x = next_pcm()
            want = a.update(x)
            if len(pending) >= 3:                         # (at most 3 in flight with results we still hold: the 4 pinned slots are reused round-robin)
                b.wait(); settle()
            if rng.integers(0, 2):
                pins[slot][...] = x
                out = b.update_async(pins[slot], pouts[slot])
                slot = (slot + 1) % 4
            else:
                out = b.update_async(x)
            pending.append((out, want))
        elif op == 5:
            b.wait(); settle()
        elif op == 6:                                     # a synchronous update in between: drains, then runs
            x = next_pcm()
            want = a.update(x)
            got = b.update(x)
            settle()
            assert np.array_equal(got, want), ('sync update after async ones differs', n, chunk, kw)
        elif op == 7:
            fa, fb = a.get_vectors(), b.get_vectors()
            settle()
            assert np.array_equal(fa, fb), ('feature windows differ', n, chunk, kw)
            if rng.integers(0, 2):
                assert np.array_equal(a.predict(fa), b.predict(fb))
        elif op == 8 and can_many:
            xs = np.stack([next_pcm() for _ in range(depth)])
            want = a.update_many(xs)
            got = b.update_many(xs)
            settle()
            assert np.array_equal(got, want), ('update_many after async updates differs', n, chunk, kw)
        elif op == 9 and rng.integers(0, 4) == 0:
            b.wait(); settle()
            a.clear(); b.clear()
        elif op == 9 and rng.integers(0, 2) == 0:         # (round 6) a masked clear: the other streams' kept leftovers must survive it
            mask = rng.random(n) < 0.3
            a.clear(mask); b.clear(mask)
            settle()
        elif op == 9:                                     # (round 6) every stream through the id list, shuffled, after kept-leftover updates
            x = next_pcm()
            want = a.update(x)
            ids = rng.permutation(n).astype(np.int32)
            got = np.empty(n, np.float32)
            got[ids] = b.update_subset(ids, x[ids])
            settle()
            assert np.array_equal(got, want), ('subset update after async ones differs', n, chunk, kw)
    b.wait(); settle()
    qa, qb = a.stream_state(), b.stream_state()
    assert all(np.array_equal(x, y) for x, y in zip(qa, qb)), ('stream state differs', n, chunk, kw)
    a.close(); b.close()
    rounds += 1
print('%d engine pairs, %d operations in %.0f s (seed %d): the pipelined engine never differs from the synchronous one' % (rounds, ops, time.time() - t0, seed))
