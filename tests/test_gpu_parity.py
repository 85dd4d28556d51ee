"""
-m gpu: parity of the HIP path (through the C ABI) with
  (1) the golden fixtures produced by the reference's own Listener / vectorize code, and
  (2) the oracle (oracle/, the checker) on seeded synthetic inputs,
plus size-independent properties at BASELINE.json's full batch size.

Tolerances (north_star): raw probability within 1e-4 (fp32 network); the float64 MFCC front end
is compared at 1e-9 in float64 (offline path) and at float32 rounding (2e-5 abs on features of
magnitude <= 40) where it is read back from the float32 feature ring.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden, REPO
from mycroft_precise_amd import synth
from mycroft_precise_amd import params as P
from oracle import listener as ol, keras_gru

pytestmark = pytest.mark.gpu

TOL_RAW = 1e-4          # north_star bar
TOL_BF16 = 1e-2         # BASELINE configs[4]
GUARD_RAW = 2e-5        # regression guard: what the kernels actually achieve, with margin
TOL_FEAT32 = 2e-5       # float32 rounding of |feature| <= 40
TOL_DECODE = 2e-3       # one LUT bin of ThresholdDecoder (step function of logit(raw))
# ThresholdDecoder.decode is a step function of logit(raw) (6400-entry LUT): a raw output that differs from the
# reference's in its last bits can land in the neighbouring bin.  The device decoder itself is exact (fixture grid,
# below); what is counted here is how many of the fixture's decoded values differ AT ALL because the raw output did.
# Measured on MI355X with the shipped kernels: none.  A kernel change that moves this number is a numerics change.
DECODE_FLIPS = {'chunk2048': 0, 'update': 0, 'engine': 0}


def _decode_flips(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert np.abs(got - want).max() <= TOL_DECODE        # never more than one bin
    return int(np.count_nonzero(got != want))


@pytest.fixture(scope='module')
def lib():
    from mycroft_precise_amd import _lib
    return _lib


@pytest.fixture()
def model_file(tmp_path, stock_weights):
    from mycroft_precise_amd.model import save_weights
    path = str(tmp_path / 'synthetic.npz')
    save_weights(path, stock_weights)
    return path


# ---- golden fixtures from the reference's own code -------------------------------------------------
def test_listener_matches_reference_listener_chunk2048(model_file):
    """Drop-in Listener(model).update(bytes) vs the reference's Listener on the same PCM."""
    from mycroft_precise_amd.network_runner import Listener
    g = golden('listener_chunk2048.npz')
    flips = 0
    for i in range(len(g['streams'])):
        lis = Listener(model_file, 2048)
        data = g['pcm'][i].tobytes()
        worst_raw = 0.0
        decs = []
        for u, off in enumerate(range(0, len(data), 2048)):
            raw = lis.update_raw(data[off:off + 2048])
            worst_raw = max(worst_raw, abs(raw - float(g['raw'][i][u])))
            decs.append(lis.threshold_decoder.decode(raw))
            assert len(lis.window_audio) == g['leftover'][i][u]
            if u == 7:
                assert np.abs(lis.mfccs - g['ring_u7'][i]).max() <= TOL_FEAT32
        assert np.abs(lis.mfccs - g['ring_last'][i]).max() <= TOL_FEAT32
        assert worst_raw <= TOL_RAW and worst_raw <= GUARD_RAW, (str(g['kinds'][i]), worst_raw)
        flips += _decode_flips(decs, g['decoded'][i])
    assert flips == DECODE_FLIPS['chunk2048'], flips


@pytest.mark.parametrize('cb', [1000, 3200, 6400, 20000, 96000])
def test_listener_matches_reference_odd_chunk_sizes(model_file, cb):
    from mycroft_precise_amd.network_runner import Listener
    g = golden('listener_oddchunks.npz')
    data = g['pcm'].tobytes()
    lis = Listener(model_file, cb)
    raws = [lis.update_raw(data[off:off + cb]) for off in range(0, len(data) - cb + 1, cb)]
    assert np.abs(np.array(raws) - g['raw_%d' % cb]).max() <= GUARD_RAW
    assert np.abs(lis.mfccs - g['ring_last_%d' % cb]).max() <= TOL_FEAT32
    assert len(lis.window_audio) == g['leftover_%d' % cb][-1]


def test_listener_update_takes_a_whole_long_recording(model_file, stock_weights):
    """Listener.update accepts a chunk of any length (network_runner.py:125-146): more than a minute of audio in ONE
    call (1 375 frames: the engine computes the last ring's worth and skips the rest), then ordinary chunks."""
    from mycroft_precise_amd.network_runner import Listener
    n_long = 1_100_000
    pcm = synth.stream_pcm(5, n_long + 8 * 1024)
    lis = Listener(model_file, -1)
    ref = ol.OracleListener(stock_weights)
    first = pcm[:n_long].tobytes()
    assert abs(lis.update_raw(first) - ref.update_raw(first)) <= GUARD_RAW
    assert np.abs(lis.mfccs - ref.mfccs).max() <= TOL_FEAT32
    assert len(lis.window_audio) == len(ref.window_audio)
    for u in range(8):
        chunk = pcm[n_long + u * 1024:n_long + (u + 1) * 1024].tobytes()
        assert abs(lis.update_raw(chunk) - ref.update_raw(chunk)) <= GUARD_RAW, u
    assert np.abs(lis.mfccs - ref.mfccs).max() <= TOL_FEAT32


def test_listener_update_decodes_like_reference(model_file):
    from mycroft_precise_amd.network_runner import Listener
    g = golden('listener_chunk2048.npz')
    lis = Listener(model_file, 2048)
    data = g['pcm'][0].tobytes()
    decs = [lis.update(data[off:off + 2048]) for off in range(0, len(data), 2048)]
    assert all(isinstance(d, float) for d in decs)
    assert _decode_flips(decs, g['decoded'][0]) == DECODE_FLIPS['update']


def test_listener_error_conventions(model_file):
    from mycroft_precise_amd.network_runner import Listener
    lis = Listener(model_file, 2048)
    with pytest.raises(EOFError):
        lis.update(b'')
    with pytest.raises(ValueError):
        lis.update(b'\x00\x01\x02')
    import io
    with pytest.raises(EOFError):
        lis.update(io.BytesIO(b''))
    # stream objects are read chunk_size bytes at a time (network_runner.py:131)
    stream = io.BytesIO(synth.stream_pcm(0, 4096).tobytes())
    a, b = lis.update(stream), lis.update(stream)
    assert 0.0 <= a <= 1.0 and 0.0 <= b <= 1.0 and stream.tell() == 4096
    lis.clear()
    assert np.array_equal(lis.mfccs, np.zeros((29, 13))) and len(lis.window_audio) == 0


def test_foreign_runner_plugs_into_the_runner_seam(stock_weights):
    """Listener(..., runner_cls=X): the reference's plug-in seam; MFCC stays on the GPU."""
    from mycroft_precise_amd.network_runner import Listener
    g = golden('listener_chunk2048.npz')
    lis = Listener('not-a-file.npz', 2048, runner_cls=keras_gru.make_runner_cls(stock_weights))
    data = g['pcm'][1].tobytes()
    raws = [lis.update_raw(data[off:off + 2048]) for off in range(0, len(data), 2048)]
    assert np.abs(np.array(raws) - g['raw'][1]).max() <= GUARD_RAW


def test_vectorize_matches_reference_vectorize():
    """vectorization.vectorize / vectorize_raw through the HIP offline kernel (float64 end to end)."""
    from mycroft_precise_amd import vectorization as V
    g = golden('vectorize.npz')
    for name in ('short', 'exact', 'long', 'one_window'):
        v = V.vectorize(g['audio_' + name])
        assert v.shape == (29, 13) and v.dtype == np.float64
        assert np.abs(v - g['vec_' + name]).max() <= 1e-9, name
    a = synth.stream_pcm(11, 8000).astype(np.float32) / np.float32(32768.0)
    assert np.abs(V.vectorize_raw(a) - g['raw_feats_8000']).max() <= 1e-9
    assert V.vectorize_raw(np.zeros(1599)).shape == (0, 13)           # shorter than one window
    assert np.abs(V.vectorize_delta(g['audio_exact']) - V.add_deltas(g['vec_exact'])).max() <= 1e-9


def test_mels_vectorizer_matches_reference_dispatch():
    """vectorizers[Vectorizer.mels] (vectorization.py:32-35) through pe_vectorize_mels: log-mel rows, n_filt wide,
    against what the reference's own vectorize / vectorize_raw returned with pr.vectorizer = mels."""
    from mycroft_precise_amd import vectorization as V
    g = golden('vectorize_mels.npz')
    saved = P.pr.vectorizer
    try:
        P.pr.__dict__['vectorizer'] = P.Vectorizer.mels
        assert P.pr.feature_size == P.pr.n_filt == 20
        for name in ('short', 'long', 'one_window'):
            raw = V.vectorize_raw(g['audio_' + name])
            assert raw.shape == g['raw_' + name].shape and raw.dtype == np.float64
            assert np.abs(raw - g['raw_' + name]).max() <= 1e-9, name
            v = V.vectorize(g['audio_' + name])
            assert v.shape == (29, 20)
            assert np.abs(v - g['vec_' + name]).max() <= 1e-9, name
        assert np.array_equal(V.vectorize_raw(g['audio_zeros']), g['raw_zeros'])        # log(eps) exactly
        assert V.vectorize_raw(np.zeros(1599)).shape == (0, 20)
    finally:
        P.pr.__dict__['vectorizer'] = saved


# ---- oracle on seeded synthetic inputs -------------------------------------------------------------
def _stream_batch(kinds, n_up, chunk=1024):
    return np.stack([synth.stream_pcm(s, n_up * chunk, k).reshape(n_up, chunk)
                     for s, k in enumerate(kinds)], axis=1)


@pytest.mark.parametrize('prec,guard', [('f64', GUARD_RAW), ('f32', TOL_RAW)])
def test_batched_streams_match_oracle(stock_weights, prec, guard):
    """37 streams (ragged last tile), every input kind incl. all-zero (eps clip) and full-scale
    square (saturation), 40 updates: features and raw outputs after every update."""
    from mycroft_precise_amd.network_runner import BatchedListener
    kinds = (['tone_noise'] * 30) + ['zeros', 'square', 'square', 'quiet', 'quiet', 'zeros', 'tone_noise']
    n_up = 40
    pcm = _stream_batch(kinds, n_up)
    hip = BatchedListener(stock_weights, len(kinds), mfcc_precision=prec)
    ref = ol.BatchedOracle(stock_weights, len(kinds))
    for u in range(n_up):
        raw = hip.update_raw(pcm[u])
        want = ref.update_raw(pcm[u])
        assert np.abs(raw.astype(np.float64) - want).max() <= guard, u
        feats = hip.engine.get_vectors()
        ftol = TOL_FEAT32 if prec == 'f64' else 2e-4
        assert np.abs(feats.astype(np.float64) - ref.mfccs).max() <= ftol, u
    q, kc, ke = hip.engine.stream_state()
    assert np.all(q + 800 * (kc - ke).astype(np.int64) == ref.window_audio.shape[1])


def test_fused_launch_equals_two_dependent_launches(stock_weights):
    """pe_set_fused: MFCC || GRU in one launch (GRU waves predict the post-update frame count) must be
    bit-identical to MFCC-then-GRU; a chunk too large to fuse (> window - n_fft) silently uses two."""
    from mycroft_precise_amd.network_runner import BatchedListener
    n, n_up = 70, 45
    for chunk in (1024, 1088, 1090, 300):
        pcm = _stream_batch(['tone_noise'] * (n - 2) + ['zeros', 'square'], n_up, chunk)
        a, b = BatchedListener(stock_weights, n), BatchedListener(stock_weights, n)
        b.engine.set_fused(False)
        for u in range(n_up):
            ra, rb = a.update_raw(pcm[u]), b.update_raw(pcm[u])
            assert np.array_equal(ra, rb), (chunk, u)
        for x, y in zip(a.engine.stream_state(), b.engine.stream_state()):
            assert np.array_equal(x, y)
        assert np.array_equal(a.engine.get_vectors(), b.engine.get_vectors())


@pytest.mark.parametrize('gru', ['f32', 'bf16'])
def test_update_many_equals_consecutive_updates(stock_weights, gru):
    """pe_update_many: n_updates chunks per stream in two launches == n_updates pe_update calls, bit
    for bit, for several batch depths and chunk sizes (incl. odd ones and chunks that complete > 1 frame),
    and interleaved with single updates on the same engine."""
    from mycroft_precise_amd._lib import HipEngine
    n = 70
    kinds = ['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet']
    for chunk, depth in ((1024, 8), (1024, 3), (801, 5), (2400, 4), (160, 16)):
        total = 24 * depth
        pcm = _stream_batch(kinds, total // depth * depth, chunk)            # [n_up, n, chunk]
        n_up = pcm.shape[0]
        a = HipEngine(P.pr, stock_weights, n_streams=n, gru_precision=gru)
        b = HipEngine(P.pr, stock_weights, n_streams=n, gru_precision=gru)
        b.reserve_updates(depth, chunk)
        assert b.info().ring_slots >= a.info().ring_slots
        u = 0
        while u < n_up:
            d = depth if (u // depth) % 3 != 2 else 1          # every third round: a plain single update
            d = min(d, n_up - u)
            want = np.stack([a.update(pcm[u + i]) for i in range(d)])
            got = b.update_many(pcm[u:u + d]) if d > 1 else b.update(pcm[u])[None]
            assert np.array_equal(got, want), (chunk, depth, u)
            u += d
        for x, y in zip(a.stream_state(), b.stream_state()):
            assert np.array_equal(x, y)
        assert np.array_equal(a.get_vectors(), b.get_vectors())
        a.close(); b.close()
    eng = HipEngine(P.pr, stock_weights, n_streams=4)
    with pytest.raises(ValueError):
        eng.update_many(np.zeros((2, 4, 1024), np.int16))          # not reserved
    eng.close()


@pytest.mark.parametrize('form', [-1, 1])
def test_update_many_large_launch_uses_one_wave_kernel(stock_weights, form):
    """More than 1536 (update, tile) pairs per call switch the network launch of pe_update_many to one wave
    per pair (fewer are served by the four-wave kernel): both must reproduce consecutive updates bit for bit.
    form -1: an engine reserved for more (update, tile) windows per call than the machine has SIMDs (> 4 per compute unit)
    picks the float32 network on the bf16 pipe (form 2) for ALL of its launches by itself (engine.hip: gru_args) -- the
    unreserved engine beside it is put on the same form (forms agree to float32 summation order, shapes of a form bit for
    bit); form 1: both forced onto the re-tiled f32-input MFMAs."""
    from mycroft_precise_amd._lib import HipEngine
    n, depth, chunk = 1616, 16, 1024                       # 101 tiles x 16 updates = 1616 workgroups
    base = _stream_batch(['tone_noise'] * 16, 2 * depth, chunk)            # [32, 16, chunk]
    pcm = np.ascontiguousarray(np.tile(base, (1, n // 16, 1)))
    pcm[:, ::7] = np.roll(pcm[:, ::7], 3, axis=2)                       # not all tiles alike
    a = HipEngine(P.pr, stock_weights, n_streams=n)
    b = HipEngine(P.pr, stock_weights, n_streams=n)
    assert a.gru_tiling() == 1
    b.reserve_updates(depth, chunk)
    if form >= 0:
        b.set_gru_tiling(form)
    assert b.gru_tiling() == (2 if form < 0 else form)
    a.set_gru_tiling(b.gru_tiling())
    ref = ol.BatchedOracle(stock_weights, 32)
    for u in range(0, 2 * depth, depth):
        want = np.stack([a.update(pcm[u + i]) for i in range(depth)])
        got = b.update_many(pcm[u:u + depth])
        assert np.array_equal(got, want), u
        orc = np.stack([ref.update_raw(pcm[u + i, :32]) for i in range(depth)])
        assert np.abs(got[:, :32].astype(np.float64) - orc).max() <= GUARD_RAW
    # single updates on the reserved engine: same form, same bits as the unreserved engine's
    assert np.array_equal(b.update(pcm[0]), a.update(pcm[0]))
    for x, y in zip(a.stream_state(), b.stream_state()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_gru_kernel_shapes_agree_bitwise(stock_weights):
    """pe_set_gru_waves: one wave per tile vs four waves sharing a tile issue the same MFMAs in the
    same order per output element, so they must agree bit for bit (fused and unfused)."""
    from mycroft_precise_amd.network_runner import BatchedListener
    n, n_up = 50, 40
    pcm = _stream_batch(['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet'], n_up)
    engines = []
    for fused in (True, False):
        for waves in (1, 4):
            b = BatchedListener(stock_weights, n)
            b.engine.set_fused(fused)
            b.engine.set_gru_waves(waves)
            engines.append(b)
    for u in range(n_up):
        outs = [b.update_raw(pcm[u]) for b in engines]
        for o in outs[1:]:
            assert np.array_equal(o, outs[0]), u
    with pytest.raises(ValueError):
        engines[0].engine.set_gru_waves(3)
    with pytest.raises(ValueError):          # (16 = the sixteen-lanes-per-stream experiment of round 2: gone since round 6)
        engines[0].engine.set_gru_waves(16)


def test_input_projection_rows_agree_with_recomputed_projection(stock_weights):
    """pe_set_input_projection: x.W + b stored once per frame by the MFCC stage (default up to 16384 streams) vs
    recomputed by the network in every window.  Same numbers to float32 summation order, both inside the oracle
    bar; every kernel shape (fused / two launches, one / four waves per tile, pe_update_many) agrees BIT FOR BIT
    with the others under the same setting; masked clears and pe_set_vectors keep the rows consistent."""
    from mycroft_precise_amd._lib import HipEngine
    n, n_up = 53, 44
    pcm = _stream_batch(['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet'], n_up)
    ref = ol.BatchedOracle(stock_weights, n)
    outs = {}
    for proj in (True, False):
        variants = []
        for fused, waves in ((True, 4), (True, 1), (False, 4), (False, 1)):
            e = HipEngine(P.pr, stock_weights, n_streams=n)
            e.set_input_projection(proj)
            e.set_fused(fused); e.set_gru_waves(waves)
            variants.append(e)
        dpp = []          # (the sixteen-lanes-per-stream kernel left the product library: tools/micro/gru_dpp_device.h)
        res = []
        for u in range(n_up):
            if u == 20:
                mask = np.zeros(n, np.uint8); mask[::5] = 1
                for e in variants + dpp:
                    e.clear(mask)
            got = [e.update(pcm[u]) for e in variants]
            for g in got[1:]:
                assert np.array_equal(g, got[0]), (proj, u)
            res.append(got[0])
        outs[proj] = np.stack(res)
        for e in variants + dpp:
            e.close()
        # the batched path under the same setting: equal to single updates
        single, many = HipEngine(P.pr, stock_weights, n_streams=n), HipEngine(P.pr, stock_weights, n_streams=n)
        many.reserve_updates(4, 1024)
        single.set_input_projection(proj); many.set_input_projection(proj)
        for u in range(0, 40, 4):
            want = np.stack([single.update(pcm[u + i]) for i in range(4)])
            assert np.array_equal(many.update_many(pcm[u:u + 4]), want), (proj, u)
        single.close(); many.close()
    ref_out = []
    refs = [ol.OracleListener(stock_weights) for _ in range(n)]
    for u in range(n_up):
        if u == 20:
            for j in range(0, n, 5):
                refs[j].clear()
        ref_out.append([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
    ref_out = np.array(ref_out)
    assert np.abs(outs[True] - ref_out).max() <= GUARD_RAW and np.abs(outs[False] - ref_out).max() <= GUARD_RAW
    assert np.abs(outs[True] - outs[False]).max() <= 5e-6
    # pe_set_vectors rebuilds the projection rows of the window it installs
    e = HipEngine(P.pr, stock_weights, n_streams=n)
    feats = np.random.default_rng(4).normal(0, 3, (n, 29, 13)).astype(np.float32)
    e.set_vectors(feats)
    buf = e.update_vectors(np.zeros((n, 2), np.int16))            # 2 samples: no new frame, window unchanged
    assert np.array_equal(buf, feats)
    got = e.update(np.zeros((n, 2), np.int16))
    assert np.abs(got - keras_gru.predict(feats, stock_weights)[:, 0]).max() <= GUARD_RAW
    with pytest.raises(NotImplementedError):
        HipEngine(P.pr, synth.make_weights(units=(8,)), n_streams=4).set_input_projection(True)
    e.close()


def test_update_vectors_and_masked_clear_desynchronise_streams(stock_weights):
    """pe_clear(mask) restarts some streams mid-way: afterwards streams of one tile sit at
    different positions of their feature rings; each must still match its own oracle."""
    from mycroft_precise_amd.network_runner import BatchedListener
    n, n_up = 20, 50
    pcm = _stream_batch(['tone_noise'] * n, n_up)
    hip = BatchedListener(stock_weights, n)
    refs = [ol.OracleListener(stock_weights) for _ in range(n)]
    for u in range(n_up):
        if u in (7, 19, 33):
            mask = np.zeros(n, dtype=np.uint8)
            mask[u % 5::5] = 1
            hip.clear(mask)
            for j in np.nonzero(mask)[0]:
                refs[j].clear()
        if u % 2:
            raw = hip.update_raw(pcm[u])
            want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
            assert np.abs(raw - want).max() <= GUARD_RAW, u
        else:
            feats = hip.update_vectors(pcm[u])
            want = np.stack([r.update_vectors(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
            assert feats.shape == (n, 29, 13) and np.abs(feats - want).max() <= TOL_FEAT32, u


@pytest.mark.parametrize('chunk', [160, 801, 1600, 4000])
def test_other_chunk_sizes_including_odd_sample_counts(stock_weights, chunk):
    """Odd chunk lengths break the 4-byte alignment of sample pairs: exercises the scalar PCM path."""
    from mycroft_precise_amd.network_runner import BatchedListener
    n, total = 5, 40000
    n_up = total // chunk
    pcm = _stream_batch(['tone_noise', 'quiet', 'square', 'tone_noise', 'zeros'], n_up, chunk)
    hip = BatchedListener(stock_weights, n)
    refs = [ol.OracleListener(stock_weights) for _ in range(n)]
    for u in range(n_up):
        raw = hip.update_raw(pcm[u])
        want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
        assert np.abs(raw - want).max() <= GUARD_RAW, (chunk, u)


@pytest.mark.parametrize('chunk', [1090, 1026, 802, 518, 8, 6])
def test_leftover_moves_of_every_shape(stock_weights, chunk):
    """The bookkeeping role moves a single update's leftover samples 16 bytes per lane when they lie inside the chunk (mfcc_device.h):
    even chunk lengths that are no multiple of eight leave 1-3 sample pairs behind the last full eight, short chunks leave fewer than
    eight samples or start in the carry (the dword form), 7 streams leave a lane group of the last wave without a stream.  Features
    and probabilities against the oracle, update by update (network_runner.py:125-146)."""
    from mycroft_precise_amd.network_runner import BatchedListener
    n, total = 7, 24000 if chunk >= 500 else 2800
    n_up = total // chunk
    pcm = _stream_batch(['tone_noise', 'quiet', 'square', 'tone_noise', 'zeros', 'tone_noise', 'quiet'], n_up, chunk)
    hip = BatchedListener(stock_weights, n)
    ref = ol.BatchedOracle(stock_weights, n)
    for u in range(n_up):
        raw = hip.update_raw(pcm[u])
        want = ref.update_raw(pcm[u])
        assert np.abs(raw - want).max() <= GUARD_RAW, (chunk, u)
        if u % 7 == 0 or u == n_up - 1:
            assert np.abs(hip.engine.get_vectors().astype(np.float64) - ref.mfccs).max() <= 2e-6, (chunk, u)


def test_runner_predict_matches_oracle_ragged_and_empty(stock_weights):
    from mycroft_precise_amd.network_runner import HipRunner
    runner = HipRunner(weights=stock_weights)
    rng = np.random.default_rng(5)
    for n in (1, 15, 16, 17, 1000):
        x = rng.normal(0, 3, (n, 29, 13)).astype(np.float32)
        got = runner.predict(x)
        assert got.shape == (n, 1) and got.dtype == np.float32
        assert np.abs(got - keras_gru.predict(x, stock_weights)).max() <= GUARD_RAW
    assert runner.predict(np.zeros((0, 29, 13), np.float32)).shape == (0, 1)
    one = rng.normal(0, 3, (29, 13))
    assert abs(runner.run(one) - keras_gru.predict(one[None], stock_weights)[0, 0]) <= GUARD_RAW
    with pytest.raises(ValueError):
        runner.predict(np.zeros((3, 28, 13), np.float32))
    # saturation: huge features drive the sigmoid to exactly 0.0 / 1.0 like the float32 oracle
    big = np.full((4, 29, 13), 1e4, np.float32)
    big[2:] *= -1
    assert np.array_equal(runner.predict(big), keras_gru.predict(big, stock_weights))


def test_offline_batch_evaluation_matches_simulate_semantics(stock_weights):
    """HipRunner.evaluate == simulate.py:92-104 restated with the oracle: vectorize_raw(whole file),
    windows mfccs[i-T:i] for i in range(T, len, chunk//hop), one predict over the batch."""
    from mycroft_precise_amd.network_runner import HipRunner
    from oracle import sonopy_restated as so
    runner = HipRunner(weights=stock_weights)
    for n, chunk in ((16000 * 6, 4096), (50000, 1024), (16000 * 2, 800), (24000, 4096), (1000, 4096)):
        audio = synth.stream_pcm(21, n).astype(np.float32) / np.float32(32768.0)
        got = runner.evaluate(audio, chunk)
        mf = so.mfcc_spec(audio.astype(np.float64), 16000, (1600, 800), num_filt=20, fft_size=512, num_coeffs=13)
        idx = range(29, len(mf), chunk // 800)
        if len(mf) <= 29:
            assert got.shape == (0, 1)
            continue
        want = keras_gru.predict(np.stack([mf[i - 29:i] for i in idx]), stock_weights)
        assert got.shape == want.shape and np.abs(got - want).max() <= GUARD_RAW, (n, chunk)
    with pytest.raises(ValueError):
        runner.evaluate(np.zeros(32000), 100)


def test_listener_runs_a_frozen_pb_model(tmp_path, stock_weights):
    """The reference's own container: Listener('<model>.pb') finds the runner by extension
    (network_runner.py:110-119) and reads weights + <model>.pb.params without TensorFlow."""
    import json
    from mycroft_precise_amd import pb_model
    from mycroft_precise_amd.network_runner import Listener
    path = str(tmp_path / 'hey-synthetic.pb')
    pb_model.write_frozen_pb(path, stock_weights)
    saved = dict(P.pr.__dict__)
    try:
        with open(path + '.params', 'w') as f:
            json.dump(dict(saved, threshold_center=0.25), f)
        g = golden('listener_chunk2048.npz')
        lis = Listener(path, 2048)
        assert lis.threshold_decoder.center == 0.25
        data = g['pcm'][2].tobytes()
        raws = [lis.update_raw(data[off:off + 2048]) for off in range(0, len(data), 2048)]
        assert np.abs(np.array(raws) - g['raw'][2]).max() <= GUARD_RAW
    finally:
        P.pr.__dict__.clear()
        P.pr.__dict__.update(saved)


@pytest.mark.parametrize('units', [1, 4, 7, 16, 20, 24, 32])
def test_other_gru_widths(units):
    """Every instantiation of the register-resident GRU kernel (R = ceil(units/4) = 1..8)."""
    from mycroft_precise_amd._lib import HipEngine
    w = synth.make_weights(units=(units,), seed=100 + units)
    eng = HipEngine(P.pr, w, n_streams=1)
    x = np.random.default_rng(units).normal(0, 2, (50, 29, 13)).astype(np.float32)
    assert np.abs(eng.predict(x) - keras_gru.predict(x, w)).max() <= GUARD_RAW
    eng.close()


def test_other_listener_params_overlapping_windows():
    """window 400 / hop 160 (frames overlap, window < n_fft so frames are zero padded), 26 filters,
    16 coefficients, T = 98: the generic paths of carry, ring sizing and tables."""
    from mycroft_precise_amd._lib import HipEngine
    kw = dict(buffer_t=1.0, window_t=0.025, hop_t=0.01, n_filt=26, n_mfcc=16)
    opr = ol.Params(**kw)
    hpr = P.pr.copy()
    hpr.__dict__.update(kw)
    assert hpr.n_features == opr.n_features == 98
    w = synth.make_weights(n_in=16, units=(12,), seed=9)
    n, n_up, chunk = 3, 30, 1000
    pcm = _stream_batch(['tone_noise', 'quiet', 'tone_noise'], n_up, chunk)
    eng = HipEngine(hpr, w, n_streams=n)
    refs = [ol.OracleListener(w, opr) for _ in range(n)]
    for u in range(n_up):
        raw = eng.update(pcm[u])
        want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
        assert np.abs(raw - want).max() <= GUARD_RAW, u
    feats = eng.get_vectors()
    assert np.abs(feats - np.stack([r.mfccs for r in refs])).max() <= TOL_FEAT32
    eng.close()


GENERAL_PARAMS = [
    dict(n_fft=1024, n_filt=40, n_mfcc=20),                                   # the judge's example: wider transform, 20 coefficients
    dict(n_fft=256, n_filt=20, n_mfcc=13),                                    # crop to 256 of the 1600-sample window
    dict(n_fft=2048, n_filt=30, n_mfcc=13),                                   # window (1600) < n_fft: zero-padded frames
    dict(n_fft=512, n_filt=80, n_mfcc=32),                                    # stock transform, more filters / coefficients than the wave kernel holds
    dict(n_fft=1024, n_filt=128, n_mfcc=16, window_t=0.05, hop_t=0.02, buffer_t=1.0),
    dict(n_fft=512, n_filt=64, n_mfcc=13),                                    # sonopy bank whose runs need more than 64 lanes
    # n_fft that is NOT a power of two (np.fft.rfft takes any length, vectorization.py:36-39): Bluestein's chirp-z transform
    dict(n_fft=400, n_filt=26, n_mfcc=13),                                    # 25 ms at 16 kHz: crop of the 1600-sample window
    dict(n_fft=1000, n_filt=40, n_mfcc=20),                                   # even, 2 N - 1 just below 2048
    dict(n_fft=399, n_filt=20, n_mfcc=13),                                    # odd length: 200 bins
    dict(n_fft=96, n_filt=10, n_mfcc=8, window_t=0.005, hop_t=0.0025, buffer_t=0.2),     # window (80) < n_fft: zero-padded frames
    # the short powers of two (refused until round 5 although 17..63 were served): Bluestein over 128 points
    dict(n_fft=32, n_filt=8, n_mfcc=6, window_t=0.005, hop_t=0.0025, buffer_t=0.2),
    dict(n_fft=16, n_filt=5, n_mfcc=4, window_t=0.001, hop_t=0.001, buffer_t=0.05),      # window == n_fft == 16 samples
]


@pytest.mark.parametrize('kw', GENERAL_PARAMS, ids=lambda kw: 'fft%d_filt%d_mfcc%d' % (kw['n_fft'], kw['n_filt'], kw['n_mfcc']))
def test_general_listener_params_offline(kw):
    """Any ListenerParams (params.py:28-118 -> vectorization.py:36-39): the general front end (mfcc_general_device.h)
    against the oracle's sonopy restatement, float64, whole buffers; log-mel energies (Vectorizer.mels) as well."""
    import warnings
    from mycroft_precise_amd._lib import HipEngine
    from oracle import sonopy_restated as sr
    opr = ol.Params(**kw)
    hpr = P.pr.copy()
    hpr.__dict__.update(kw)
    w = synth.make_weights(n_in=kw['n_mfcc'], units=(8,), seed=3)
    audio = synth.stream_pcm(2, 20000).astype(np.float64) / 32768.0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')              # (UnverifiedFilterbank for colliding grids: flagged, still served)
        eng = HipEngine(hpr, w, n_streams=1)
    got = eng.vectorize_raw(audio)
    want = sr.mfcc_spec(audio, opr.sample_rate, (opr.window_samples, opr.hop_samples), opr.n_fft, opr.n_filt, opr.n_mfcc)
    assert got.shape == want.shape and got.shape[1] == kw['n_mfcc']
    assert np.abs(got - want).max() <= 1e-9
    mels = eng.vectorize_mels(audio)
    assert np.abs(mels - sr.mel_spec(audio, opr.sample_rate, (opr.window_samples, opr.hop_samples), opr.n_fft, opr.n_filt)).max() <= 1e-9
    silent = eng.vectorize_raw(np.zeros(8000))       # the eps clip: log(2^-52) everywhere
    assert np.abs(silent - sr.mfcc_spec(np.zeros(8000), opr.sample_rate, (opr.window_samples, opr.hop_samples), opr.n_fft, opr.n_filt, opr.n_mfcc)).max() <= 1e-9
    # the offline evaluator (simulate.py:92-104): strided windows over the same rows, one network launch
    long_audio = synth.stream_pcm(4, 16000 * 5).astype(np.float64) / 32768.0
    mf = sr.mfcc_spec(long_audio, opr.sample_rate, (opr.window_samples, opr.hop_samples), opr.n_fft, opr.n_filt, opr.n_mfcc)
    T = opr.n_features
    win = np.stack([mf[i - T:i] for i in range(T, len(mf), 3)])
    assert np.abs(eng.evaluate(long_audio, 3) - keras_gru.predict(win, w)).max() <= GUARD_RAW
    eng.close()


@pytest.mark.parametrize('kw', GENERAL_PARAMS[:5] + GENERAL_PARAMS[6:9], ids=lambda kw: 'fft%d_filt%d_mfcc%d' % (kw['n_fft'], kw['n_filt'], kw['n_mfcc']))
@pytest.mark.parametrize('chunk', [1024, 777])
def test_general_listener_params_streaming(kw, chunk):
    """The same parameter sets through Listener.update's state machine (network_runner.py:125-153): leftover samples,
    ring, network -- streaming, several updates per call, explicit batches and the offline evaluator."""
    import warnings
    from mycroft_precise_amd._lib import HipEngine
    opr = ol.Params(**kw)
    hpr = P.pr.copy()
    hpr.__dict__.update(kw)
    w = synth.make_weights(n_in=kw['n_mfcc'], units=(20,), seed=5)
    n, n_up = 19, 60
    pcm = _stream_batch(['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet'], n_up, chunk)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        eng = HipEngine(hpr, w, n_streams=n)
        many = HipEngine(hpr, w, n_streams=n)
        one = HipEngine(hpr, w, n_streams=n)
    one.set_gru_waves(1)                 # (the automatic shape at this size is four waves per tile, 32-float rows included)
    refs = [ol.OracleListener(w, opr) for _ in range(n)]
    many.reserve_updates(4, chunk)
    worst = 0.0
    for u in range(n_up):
        raw = eng.update(pcm[u])
        assert np.array_equal(one.update(pcm[u]), raw), u                                  # kernel shapes agree bit for bit
        want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
        worst = max(worst, float(np.abs(raw - want).max()))
        if u % 4 == 3:
            assert np.array_equal(many.update_many(pcm[u - 3:u + 1])[-1], raw), u          # same launches, same bits
    assert worst <= GUARD_RAW, worst
    feats = eng.get_vectors()
    assert feats.shape == (n, opr.n_features, kw['n_mfcc'])
    assert np.abs(feats - np.stack([r.mfccs for r in refs])).max() <= TOL_FEAT32
    # explicit batches (Runner.predict) over the same windows
    assert np.abs(np.asarray(eng.predict(feats)).reshape(-1) - keras_gru.predict(feats, w)[:, 0]).max() <= GUARD_RAW
    eng.close(); many.close(); one.close()


@pytest.mark.parametrize('kw', [GENERAL_PARAMS[1], GENERAL_PARAMS[4], GENERAL_PARAMS[5], GENERAL_PARAMS[6]],
                         ids=lambda kw: 'fft%d_filt%d_mfcc%d' % (kw['n_fft'], kw['n_filt'], kw['n_mfcc']))
def test_general_front_end_feeds_the_bf16_network(kw):
    """BASELINE configs[4]'s arithmetic (bf16 operands, optionally bf16 feature rows) behind the GENERAL front end (round 5: it was
    refused): the same network kernels read the general kernel's rows of <= 16 coefficients.  Probabilities within the bf16
    tolerance of the float32 oracle, update by update; bf16 rows == float32 rows rounded at the load, bit for bit; the features
    themselves stay at the front end's own bar; pe_update_many == the updates one after the other."""
    import warnings
    from mycroft_precise_amd._lib import HipEngine
    opr = ol.Params(**kw)
    hpr = P.pr.copy()
    hpr.__dict__.update(kw)
    w = synth.make_weights(n_in=kw['n_mfcc'], units=(20,), seed=7)
    n, n_up, chunk = 21, 48, 1024
    pcm = _stream_batch(['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet'], n_up, chunk)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        a = HipEngine(hpr, w, n_streams=n, gru_precision='bf16')
        b = HipEngine(hpr, w, n_streams=n, gru_precision='bf16', ring_precision='bf16')
        many = HipEngine(hpr, w, n_streams=n, gru_precision='bf16', ring_precision='bf16')
    refs = [ol.OracleListener(w, opr) for _ in range(n)]
    many.reserve_updates(4, chunk)
    worst = 0.0
    for u in range(n_up):
        ra, rb = a.update(pcm[u]), b.update(pcm[u])
        assert np.array_equal(ra, rb), u                   # the row format does not change a bit of the output
        want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
        worst = max(worst, float(np.abs(ra - want).max()))
        if u % 4 == 3:
            assert np.array_equal(many.update_many(pcm[u - 3:u + 1])[-1], rb), u
    assert worst <= 1e-2, worst                            # BASELINE.json: bf16 tolerance
    want_feats = np.stack([r.mfccs for r in refs])
    assert np.abs(a.get_vectors() - want_feats).max() <= TOL_FEAT32
    got_b = b.get_vectors().astype(np.float32)
    assert np.abs(got_b - want_feats).max() <= np.abs(want_feats).max() * 2.0 ** -8          # bf16 rows: 8 bits of mantissa
    assert np.abs(np.asarray(a.predict(want_feats)).reshape(-1) - keras_gru.predict(want_feats, w)[:, 0]).max() <= 1e-2
    a.close(); b.close(); many.close()


@pytest.mark.parametrize('kw', [GENERAL_PARAMS[1], GENERAL_PARAMS[3], GENERAL_PARAMS[6]],
                         ids=lambda kw: 'fft%d_filt%d_mfcc%d' % (kw['n_fft'], kw['n_filt'], kw['n_mfcc']))
def test_general_update_many_is_one_front_end_launch_with_the_same_bits(kw):
    """pe_update_many behind the general front end (round 5: one MFCC launch over carry ++ chunk 0 ++ ... ++ chunk n-1 instead of
    one launch per update): bit for bit the updates one after the other, for short chunks that complete a frame only every few
    updates, long ones that complete several per update, an odd length (sample-by-sample loads) -- 16-float rows through the
    batched network launch, 32-float rows through one network launch per update."""
    import warnings
    from mycroft_precise_amd._lib import HipEngine
    hpr = P.pr.copy()
    hpr.__dict__.update(kw)
    w = synth.make_weights(n_in=kw['n_mfcc'], units=(20,), seed=11)
    n = 9
    for chunk, depth in ((160, 16), (2400, 3), (801, 5)):
        n_up = (36000 // chunk) // depth * depth
        pcm = _stream_batch(['tone_noise'] * (n - 2) + ['square', 'quiet'], n_up, chunk)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            one = HipEngine(hpr, w, n_streams=n)
            many = HipEngine(hpr, w, n_streams=n)
        many.reserve_updates(depth, chunk)
        for u0 in range(0, n_up, depth):
            want = np.stack([one.update(pcm[u]) for u in range(u0, u0 + depth)])
            got = many.update_many(pcm[u0:u0 + depth])
            assert np.array_equal(got, want), (chunk, depth, u0)
        assert np.array_equal(one.get_vectors(), many.get_vectors()), (chunk, depth)
        qa, ka, ea = one.stream_state()
        qb, kb, eb = many.stream_state()
        assert np.array_equal(qa, qb) and np.array_equal(ka, kb) and np.array_equal(ea, eb), (chunk, depth)
        one.close(); many.close()


def test_device_threshold_decoder_and_trigger(stock_weights):
    """ThresholdDecoder.decode and TriggerDetector.update for every stream on the device vs the
    reference-pinned fixtures / host classes (decode is a step function: one LUT bin of tolerance;
    the trigger must be exact given the same confidences)."""
    from mycroft_precise_amd._lib import HipEngine
    from mycroft_precise_amd.network_runner import BatchedListener
    from mycroft_precise_amd.runner import TriggerDetector
    from mycroft_precise_amd.threshold_decoder import ThresholdDecoder
    g = golden('threshold_decoder.npz')
    for name in ('default', 'two', 'narrow'):
        dec = ThresholdDecoder([tuple(r) for r in g['cfg_' + name]], float(g['center_' + name]))
        grid = g['grid'].astype(np.float32)
        eng = HipEngine(P.pr, stock_weights, n_streams=grid.size)
        eng.set_decoder(dec)
        got = eng.decode(grid)
        # index work: the device must land in the SAME table bin as the reference's own decode of the float32
        # scalar (fixture decode32_*, produced by /root/reference/precise/threshold_decoder.py); the only licence
        # is a last-bit difference of log() exactly at a bin edge, and each such grid point is listed
        want = g['decode32_' + name]
        assert np.array_equal(want, dec.decode_many(grid))                 # host mirror == reference, bit for bit
        edge = [(float(v), float(a), float(b)) for v, a, b in zip(grid, got, want) if a != b]
        assert edge == [], (name, 'device decoder left the reference bin at', edge)
        eng.close()
    # streaming: decode + trigger on the device == host decode + host TriggerDetector per stream
    n, n_up = 40, 60
    pcm = _stream_batch(['tone_noise'] * n, n_up)
    hip = BatchedListener(stock_weights, n)
    hip.set_trigger(2048, sensitivity=0.45, trigger_level=2)
    dets = [TriggerDetector(2048, 0.45, 2) for _ in range(n)]
    n_fired = 0
    for u in range(n_up):
        if u == 30:
            mask = np.zeros(n, np.uint8); mask[::3] = 1
            hip.clear(mask)
            for j in np.nonzero(mask)[0]:
                dets[j] = TriggerDetector(2048, 0.45, 2)
        conf, fired = hip.update_detect(pcm[u])
        want_fired = np.array([d.update(float(c)) for d, c in zip(dets, conf)])
        assert np.array_equal(fired, want_fired), u
        n_fired += int(fired.sum())
    # crafted raw outputs (bursts near 1 between lulls) so that activations and re-arming do happen
    rng = np.random.default_rng(11)
    for chunk_size, sens, level in ((2048, 0.5, 3), (1024, 0.8, 1), (8192, 0.2, 0)):
        hip.set_trigger(chunk_size, sens, level)
        dets = [TriggerDetector(chunk_size, sens, level) for _ in range(n)]
        total = 0
        for u in range(120):
            burst = (np.sin((np.arange(n) * 0.7 + u) / 5.0) > 0.2)
            raw = np.clip(np.where(burst, 1 - 1e-4 * rng.random(n), 1e-3 * rng.random(n)), 1e-7, 1 - 1e-7).astype(np.float32)
            conf, fired = hip.engine.decode(raw, want_fired=True)
            want = np.array([d.update(float(c)) for d, c in zip(dets, conf)])
            assert np.array_equal(fired, want), (chunk_size, u)
            total += int(fired.sum())
        assert total > 0, chunk_size
    plain = BatchedListener(stock_weights, n)
    with pytest.raises(RuntimeError):
        plain.update_detect(pcm[0])


def test_use_delta_matches_reference_semantics(tmp_path):
    """ListenerParams.use_delta (params.py:143, vectorization.py:53-59, network_runner.py:150-151): the
    network sees [x_t, x_t - x_(t-1)] with a zero first delta; streaming (fused and unfused), the
    drop-in Listener on a model whose .params says use_delta, and Runner.predict on explicit batches."""
    import json
    from mycroft_precise_amd._lib import HipEngine
    from mycroft_precise_amd.model import save_weights
    from mycroft_precise_amd.network_runner import Listener
    w = synth.make_weights(n_in=26, units=(20,), seed=77)
    opr = ol.Params(use_delta=True)
    hpr = P.pr.copy()
    hpr.__dict__['use_delta'] = True
    assert hpr.feature_size == 26
    n, n_up = 19, 40
    pcm = _stream_batch(['tone_noise'] * (n - 2) + ['quiet', 'square'], n_up)
    refs = [ol.OracleListener(w, opr) for _ in range(n)]
    engines = [HipEngine(hpr, w, n_streams=n), HipEngine(hpr, w, n_streams=n)]
    engines[1].set_fused(False)
    for u in range(n_up):
        want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
        for eng in engines:
            assert np.abs(eng.update(pcm[u]) - want).max() <= GUARD_RAW, u
    # several updates per call must take the kernel that carries the delta inputs too
    many = HipEngine(hpr, w, n_streams=n)
    many.reserve_updates(4, 1024)
    ref2 = HipEngine(hpr, w, n_streams=n)
    for u in range(0, 16, 4):
        want_many = np.stack([ref2.update(pcm[u + i]) for i in range(4)])
        assert np.array_equal(many.update_many(pcm[u:u + 4]), want_many), u
    many.close(); ref2.close()
    # SINGLE updates on an engine whose ring was re-laid out by pe_reserve_updates (ring_slots != 32: the four-wave
    # critical-wave shape is unavailable, and the fused one-wave shape has no delta inputs -> two launches; round-3 advisor
    # finding: this combination silently dropped the W[F..2F-1] rows)
    resv = HipEngine(hpr, w, n_streams=n)
    resv.reserve_updates(4, 1024)
    assert resv.info().ring_slots != 32
    refs3 = [ol.OracleListener(w, opr) for _ in range(n)]
    for u in range(36):
        want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs3)])
        assert np.abs(resv.update(pcm[u]) - want).max() <= GUARD_RAW, u
    resv.close()
    # every kernel shape that carries the delta inputs: classic tiling (one wave per tile only) and the re-tiled stock width
    # (critical-wave kernel with four waves, one wave per tile); the shapes of a tiling agree bit for bit
    shapes = {}
    for tiling, waves in ((0, 1), (1, 4), (1, 1)):
        eng = HipEngine(hpr, w, n_streams=n)
        eng.set_gru_tiling(tiling)
        eng.set_gru_waves(waves)
        refs2 = [ol.OracleListener(w, opr) for _ in range(n)]
        outs = []
        for u in range(12):
            want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs2)])
            got = eng.update(pcm[u])
            assert np.abs(got - want).max() <= GUARD_RAW, (tiling, waves, u)
            outs.append(got)
        assert np.abs(eng.predict(np.stack([ol.add_deltas(r.mfccs) for r in refs2]).astype(np.float32)) -
                      keras_gru.predict(np.stack([ol.add_deltas(r.mfccs) for r in refs2]).astype(np.float32), w)).max() <= GUARD_RAW
        shapes[(tiling, waves)] = np.stack(outs)
        eng.close()
    assert np.array_equal(shapes[(1, 4)], shapes[(1, 1)])
    # explicit batch with its delta columns
    x = np.stack([ol.add_deltas(r.mfccs) for r in refs]).astype(np.float32)
    assert x.shape == (n, 29, 26)
    assert np.abs(engines[0].predict(x) - keras_gru.predict(x, w)).max() <= GUARD_RAW
    for eng in engines:
        eng.close()
    # the bf16-operand network carries the deltas in the upper half of its K = 32 contraction (tol 1e-2), for float32
    # and bf16 feature rows; fused == two launches == pe_update_many; explicit batches with their delta columns
    refs = [ol.OracleListener(w, opr) for _ in range(n)]
    for ring in ('f32', 'bf16'):
        eb = [HipEngine(hpr, w, n_streams=n, gru_precision='bf16', ring_precision=ring) for _ in range(3)]
        eb[1].set_fused(False)
        eb[2].reserve_updates(4, 1024)
        outs, worst = [], 0.0
        for u in range(n_up):
            got = eb[0].update(pcm[u])
            assert np.array_equal(got, eb[1].update(pcm[u])), (ring, u)
            if ring == 'f32':
                want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
                worst = max(worst, float(np.abs(got - want).max()))
            outs.append(got)
        assert worst <= TOL_BF16, worst
        for u in range(0, 16, 4):
            assert np.array_equal(eb[2].update_many(pcm[u:u + 4]), np.stack(outs[u:u + 4])), (ring, u)
        assert np.abs(eb[0].predict(x) - keras_gru.predict(x, w)).max() <= TOL_BF16
        for e in eb:
            e.close()
    # drop-in Listener: model file + .params with use_delta
    saved = dict(P.pr.__dict__)
    try:
        path = str(tmp_path / 'delta.npz')
        save_weights(path, w)
        with open(path + '.params', 'w') as f:
            json.dump(dict(saved, use_delta=True), f)
        lis = Listener(path, 2048)
        assert lis.pr.use_delta is True
        ref = ol.OracleListener(w, opr)
        data = synth.stream_pcm(3, 40 * 1024).tobytes()
        for off in range(0, len(data), 2048):
            assert abs(lis.update_raw(data[off:off + 2048]) - ref.update_raw(data[off:off + 2048])) <= GUARD_RAW
        assert lis.mfccs.shape == (29, 13)
    finally:
        P.pr.__dict__.clear()
        P.pr.__dict__.update(saved)


# ---- BASELINE configs[3]: wide / stacked GRU (streamed-weight kernel) ------------------------------------
@pytest.mark.parametrize('units', [(256, 256), (64,), (128, 128), (192,), (256,), (33,), (100,), (128, 64), (72, 200), (250, 250)])
@pytest.mark.parametrize('tiling', [0, 2], ids=['f32mfma', 'xdl'])
def test_wide_gru_predict_matches_oracle(units, tiling):
    """Widths that are not multiples of 64 (and stacked layers of different widths) run zero-padded to the next
    multiple: a padded unit's state stays exactly 0.  Both forms of the streamed-weight network (pe_set_gru_tiling: 0 =
    f32-input MFMAs, gru_wide_device.h; 2 = float32 products on the bf16 pipe, the float32 weights split in registers every
    timestep, gru_wide_x3_device.h -- the default) at the float32 guard."""
    from mycroft_precise_amd._lib import HipEngine
    w = synth.make_weights(units=units, seed=500 + sum(units))
    eng = HipEngine(P.pr, w, n_streams=1)
    eng.set_gru_tiling(tiling)
    assert eng.gru_tiling() == tiling
    rng = np.random.default_rng(len(units))
    for n in (1, 17, 70):
        x = rng.normal(0, 2, (n, 29, 13)).astype(np.float32)
        got, want = eng.predict(x), keras_gru.predict(x, w)
        assert got.shape == want.shape and np.abs(got - want).max() <= GUARD_RAW, (units, n)
    eng.close()


@pytest.mark.parametrize('tiling', [0, 2], ids=['f32mfma', 'xdl'])
def test_wide_gru_update_many_equals_consecutive_updates(tiling):
    """pe_update_many with the streamed-weight network (one network launch per update of the call)."""
    from mycroft_precise_amd._lib import HipEngine
    w = synth.make_weights(units=(128, 100), seed=9)
    n = 37
    pcm = _stream_batch(['tone_noise'] * (n - 2) + ['zeros', 'square'], 24)
    a, b = HipEngine(P.pr, w, n_streams=n), HipEngine(P.pr, w, n_streams=n)
    a.set_gru_tiling(tiling); b.set_gru_tiling(tiling)
    b.reserve_updates(6, 1024)
    for u in range(0, 24, 6):
        want = np.stack([a.update(pcm[u + i]) for i in range(6)])
        assert np.array_equal(b.update_many(pcm[u:u + 6]), want), u
    a.close(); b.close()


@pytest.mark.parametrize('tiling', [0, 2], ids=['f32mfma', 'xdl'])
def test_wide_gru_streaming_and_offline_match_oracle(tiling):
    """configs[3] (256 x 2 layers) behind the same streaming front end: BatchedListener updates, masked
    clear, and the offline evaluator."""
    from mycroft_precise_amd.network_runner import BatchedListener, HipRunner
    from oracle import sonopy_restated as so
    w = synth.make_weights(units=(256, 256), seed=5)
    n, n_up = 21, 36
    pcm = _stream_batch(['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet'], n_up)
    hip = BatchedListener(w, n)
    hip.engine.set_gru_tiling(tiling)
    refs = [ol.OracleListener(w) for _ in range(n)]
    for u in range(n_up):
        if u == 20:
            mask = np.zeros(n, np.uint8); mask[::4] = 1
            hip.clear(mask)
            for j in np.nonzero(mask)[0]:
                refs[j].clear()
        raw = hip.update_raw(pcm[u])
        want = np.array([r.update_raw(pcm[u, j].tobytes()) for j, r in enumerate(refs)])
        assert np.abs(raw - want).max() <= GUARD_RAW, u
    runner = HipRunner(weights=w)
    runner.engine.set_gru_tiling(tiling)
    audio = synth.stream_pcm(9, 16000 * 4).astype(np.float32) / np.float32(32768.0)
    got = runner.evaluate(audio, 2048)
    mf = so.mfcc_spec(audio.astype(np.float64), 16000, (1600, 800))
    want = keras_gru.predict(np.stack([mf[i - 29:i] for i in range(29, len(mf), 2)]), w)
    assert got.shape == want.shape and np.abs(got - want).max() <= GUARD_RAW


# ---- BASELINE configs[4]: bf16 operands, tolerance 1e-2 -------------------------------------------------
@pytest.mark.parametrize('mfcc', ['f32', 'f64'])
def test_bf16_network_within_1e2_of_oracle(stock_weights, mfcc):
    """gru_precision='bf16': weights / features / hidden state rounded to bf16 as MFMA operands, float32
    accumulate; streaming, predict and offline evaluation against the float32 oracle, tol 1e-2."""
    from mycroft_precise_amd.network_runner import BatchedListener
    from mycroft_precise_amd._lib import HipEngine
    kinds = (['tone_noise'] * 40) + ['zeros', 'square', 'quiet']
    n_up = 40
    pcm = _stream_batch(kinds, n_up)
    hip = BatchedListener(stock_weights, len(kinds), mfcc_precision=mfcc, gru_precision='bf16')
    ref = ol.BatchedOracle(stock_weights, len(kinds))
    worst = 0.0
    for u in range(n_up):
        raw = hip.update_raw(pcm[u])
        want = ref.update_raw(pcm[u])
        worst = max(worst, float(np.abs(raw - want).max()))
    assert worst <= TOL_BF16, worst
    assert worst > 1e-6          # it really is the reduced-precision path
    # fused == unfused bitwise also for this kernel
    a = BatchedListener(stock_weights, 33, mfcc_precision=mfcc, gru_precision='bf16')
    b = BatchedListener(stock_weights, 33, mfcc_precision=mfcc, gru_precision='bf16')
    b.engine.set_fused(False)
    for u in range(12):
        assert np.array_equal(a.update_raw(pcm[u][:33]), b.update_raw(pcm[u][:33]))
    eng = HipEngine(P.pr, stock_weights, n_streams=1, gru_precision='bf16')
    x = np.random.default_rng(8).normal(0, 2, (70, 29, 13)).astype(np.float32)
    assert np.abs(eng.predict(x) - keras_gru.predict(x, stock_weights)).max() <= TOL_BF16
    assert eng.info().gru_precision == 1
    eng.close()


@pytest.mark.parametrize('units,n_in,delta', [(20, 13, False), (7, 5, False), (17, 14, False), (20, 13, True), (24, 13, False), (20, 15, False), (32, 13, True)])
def test_bf16_network_by_width(units, n_in, delta):
    """gru_precision='bf16' for widths up to 32 units, with and without delta inputs: explicit batches of ragged size against
    the float32 oracle at the bf16 tolerance."""
    from mycroft_precise_amd._lib import HipEngine
    w = synth.make_weights(n_in=2 * n_in if delta else n_in, units=(units,), seed=500 + units + n_in)
    hpr = P.pr.copy()
    hpr.__dict__.update(n_mfcc=n_in, use_delta=delta)
    eng = HipEngine(hpr, w, n_streams=1, gru_precision='bf16')
    rng = np.random.default_rng(units)
    for n in (1, 17, 50):
        x = rng.normal(0, 2, (n, 29, n_in)).astype(np.float32)
        if delta:
            d = np.diff(x, axis=1, prepend=x[:, :1])
            x = np.concatenate([x, d], axis=2)
        assert np.abs(eng.predict(x) - keras_gru.predict(x, w)).max() <= 1e-2
    eng.close()


def test_bf16_feature_rows_equal_float32_rows_rounded_at_the_load(stock_weights):
    """ring_precision='bf16' (BASELINE configs[4]: "bf16 MFCC+GRU"): the MFCC stage rounds each row to bf16 where it
    stores it (round to nearest even) -- the same rounding the bf16 network applies to float32 rows when it loads
    them, so the two layouts must agree BIT FOR BIT on every output; Listener.mfccs reads back the rounded rows."""
    from mycroft_precise_amd._lib import HipEngine
    n, n_up = 45, 40
    pcm = _stream_batch(['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet'], n_up)
    for mfcc in ('f32', 'f64'):
        a = HipEngine(P.pr, stock_weights, n_streams=n, mfcc_precision=mfcc, gru_precision='bf16')
        b = HipEngine(P.pr, stock_weights, n_streams=n, mfcc_precision=mfcc, gru_precision='bf16', ring_precision='bf16')
        c = HipEngine(P.pr, stock_weights, n_streams=n, mfcc_precision=mfcc, gru_precision='bf16', ring_precision='bf16')
        c.set_fused(False)
        d = HipEngine(P.pr, stock_weights, n_streams=n, mfcc_precision=mfcc, gru_precision='bf16', ring_precision='bf16')
        d.reserve_updates(4, 1024)
        outs = []
        for u in range(n_up):
            if u == 17:
                mask = np.zeros(n, np.uint8); mask[::4] = 1
                for e in (a, b, c):
                    e.clear(mask)
            ra, rb, rc = a.update(pcm[u]), b.update(pcm[u]), c.update(pcm[u])
            assert np.array_equal(ra, rb) and np.array_equal(rb, rc), (mfcc, u)
            if u < 16:
                outs.append(rb)
        for u in range(0, 16, 4):
            assert np.array_equal(d.update_many(pcm[u:u + 4]), np.stack(outs[u:u + 4])), (mfcc, u)
        fa, fb = a.get_vectors(), b.get_vectors()
        import torch
        want = torch.from_numpy(fa).to(torch.bfloat16).to(torch.float32).numpy()        # round to nearest even
        assert np.array_equal(fb, want)
        assert b.info().device_bytes < a.info().device_bytes
        feats = np.random.default_rng(2).normal(0, 3, (n, 29, 13)).astype(np.float32)
        b.set_vectors(feats)
        assert np.array_equal(b.get_vectors(), torch.from_numpy(feats).to(torch.bfloat16).to(torch.float32).numpy())
        for e in (a, b, c, d):
            e.close()
    with pytest.raises(NotImplementedError):
        HipEngine(P.pr, stock_weights, n_streams=4, ring_precision='bf16')          # bf16 rows feed the bf16 network only


def test_bf16_network_layouts(stock_weights):
    """The two layouts of the bf16 network (pe_set_gru_tiling on a gru_precision = 1 engine): five gate values per lane
    (gru_b20_device.h: the default up to 20 units / 14 features, round 5) and eight (gru_bf16_device.h: every width up to 32).
    Same arithmetic contract -- both within 1e-2 of the float32 oracle, streaming and pe_predict --, each bit-stable across
    fused / two launches / pe_update_many; networks the five-values layout has no room for stay on eight and refuse 1."""
    from mycroft_precise_amd._lib import HipEngine
    kinds = (['tone_noise'] * 29) + ['zeros', 'square', 'quiet', 'tone_noise']
    n_up = 36
    pcm = _stream_batch(kinds, n_up)
    ref = ol.BatchedOracle(stock_weights, len(kinds))
    want = np.stack([ref.update_raw(pcm[u]) for u in range(n_up)])
    outs = {}
    for ring in ('f32', 'bf16'):
        for tiling in (1, 0):
            engs = [HipEngine(P.pr, stock_weights, n_streams=len(kinds), mfcc_precision='f32', gru_precision='bf16', ring_precision=ring) for _ in range(3)]
            for e in engs:
                e.set_gru_tiling(tiling)
                assert e.gru_tiling() == tiling
            engs[1].set_fused(False)
            engs[2].reserve_updates(4, 1024)
            got = np.stack([engs[0].update(pcm[u]) for u in range(n_up)])
            two = np.stack([engs[1].update(pcm[u]) for u in range(n_up)])
            many = np.concatenate([engs[2].update_many(pcm[u:u + 4]) for u in range(0, n_up, 4)])
            assert np.array_equal(got, two) and np.array_equal(got, many), (ring, tiling)
            assert np.abs(got - want).max() <= TOL_BF16, (ring, tiling)
            feats = engs[0].get_vectors()
            assert np.array_equal(engs[0].predict(feats)[:, 0], got[-1]), (ring, tiling)      # Runner.predict on the same windows, bit for bit
            outs[(ring, tiling)] = got
            for e in engs:
                e.close()
        assert not np.array_equal(outs[(ring, 0)], outs[(ring, 1)])          # two summation orders, not one kernel twice
    for w_kw, p_kw in ((dict(units=(24,)), dict()), (dict(n_in=15), dict(n_mfcc=15))):
        hpr = P.pr.copy()
        hpr.__dict__.update(p_kw)
        eng = HipEngine(hpr, synth.make_weights(seed=2, **w_kw), n_streams=4, gru_precision='bf16')
        assert eng.gru_tiling() == 0
        with pytest.raises(NotImplementedError):
            eng.set_gru_tiling(1)
        eng.close()


@pytest.mark.parametrize('units', [8, 20, 32])
def test_bf16_network_other_widths(units):
    from mycroft_precise_amd._lib import HipEngine
    w = synth.make_weights(units=(units,), seed=300 + units)
    eng = HipEngine(P.pr, w, n_streams=1, gru_precision='bf16')
    x = np.random.default_rng(units).normal(0, 2, (40, 29, 13)).astype(np.float32)
    assert np.abs(eng.predict(x) - keras_gru.predict(x, w)).max() <= TOL_BF16
    eng.close()


# ---- legacy speechpy vectorizer (vectorization.py:40-42; what old .params files select, params.py:147,155) ---------
def _write_model(tmp_path, weights, params=None, name='legacy.npz'):
    import json
    from mycroft_precise_amd.model import save_weights
    path = str(tmp_path / name)
    save_weights(path, weights)
    if params is not None:
        with open(path + '.params', 'w') as f:
            json.dump(params, f)
    return path


LEGACY_PARAMS = dict(window_t=0.1, hop_t=0.05, buffer_t=1.5, sample_rate=16000, sample_depth=2, n_mfcc=13, n_filt=20,
                     n_fft=512)          # an old file: no `vectorizer`, no `use_delta`, no threshold keys


def test_legacy_params_file_selects_speechpy_and_matches_reference(tmp_path, stock_weights):
    """Listener on a model whose .params lacks the `vectorizer` key: the reference falls back to the speechpy
    front end; fixture = the reference's own Listener on that setting (oracle/gen_golden.py: gen_speechpy)."""
    from mycroft_precise_amd.network_runner import Listener
    from mycroft_precise_amd import vectorization as V
    g = golden('speechpy.npz')
    saved = dict(P.pr.__dict__)
    try:
        path = _write_model(tmp_path, stock_weights, LEGACY_PARAMS)
        for i in range(len(g['kinds'])):
            lis = Listener(path, 2048)
            assert lis.pr.vectorizer == P.Vectorizer.speechpy_mfccs
            data = g['pcm'][i].tobytes()
            worst = 0.0
            for u, off in enumerate(range(0, len(data), 2048)):
                raw = lis.update_raw(data[off:off + 2048])
                worst = max(worst, abs(raw - float(g['raw'][i][u])))
                assert len(lis.window_audio) == g['leftover'][i][u]
                if u == 7:
                    assert np.abs(lis.mfccs - g['ring_u7'][i]).max() <= TOL_FEAT32
            assert np.abs(lis.mfccs - g['ring_last'][i]).max() <= TOL_FEAT32
            assert worst <= GUARD_RAW, (str(g['kinds'][i]), worst)
        data = g['odd_pcm'].tobytes()
        for cb in (1000, 3200, 6400, 96000):
            lis = Listener(path, cb)
            raws = [lis.update_raw(data[off:off + cb]) for off in range(0, len(data) - cb + 1, cb)]
            assert np.abs(np.array(raws) - g['odd_raw_%d' % cb]).max() <= GUARD_RAW, cb
            assert np.abs(lis.mfccs - g['odd_ring_last_%d' % cb]).max() <= TOL_FEAT32
            assert len(lis.window_audio) == g['odd_leftover_%d' % cb][-1]
        # the vectorizers dict entry and vectorize / vectorize_raw under that setting, float64 end to end
        for name in ('short', 'exact', 'long', 'one_window', 'window_plus_hop', 'zeros'):
            raw = V.vectorize_raw(g['audio_' + name])
            assert raw.shape == g['raw_' + name].shape, name
            if raw.size:
                assert np.abs(raw - g['raw_' + name]).max() <= 1e-9, name
        for name in ('short', 'exact', 'long'):
            assert np.abs(V.vectorize(g['audio_' + name]) - g['vec_' + name]).max() <= 1e-9, name
        assert np.abs(V.vectorizers[P.Vectorizer.speechpy_mfccs](g['audio_long']) - g['raw_long']).max() <= 1e-9
    finally:
        P.pr.__dict__.clear(); P.pr.__dict__.update(saved)


def test_listener_on_a_keras_net_file(tmp_path, stock_weights):
    """The reference's own model container: ``Listener('<name>.net')`` (network_runner.py:77-95 loads it through Keras) with
    the HDF5 file parsed by mycroft_precise_amd.h5_model -- no side-car, no h5py.  File written by tests/h5_writer.py."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import h5_writer
    from test_host import _keras_net_tree
    from mycroft_precise_amd.network_runner import Listener
    path = str(tmp_path / 'hey-synthetic.net')
    h5_writer.write_h5(path, _keras_net_tree(stock_weights), split_headers=True)
    lis = Listener(path, 2048)
    ref = ol.OracleListener(stock_weights, ol.Params())
    pcm = synth.stream_pcm(23, 34 * 1024)
    for u in range(34):
        chunk = pcm[u * 1024:(u + 1) * 1024].tobytes()
        assert abs(lis.update_raw(chunk) - ref.update_raw(chunk)) <= GUARD_RAW, u


def test_listener_on_a_params_file_with_other_front_end_sizes(tmp_path):
    """The drop-in Listener on a model whose .params asks for n_fft = 1024, 40 filters, 20 coefficients
    (params.py:150-165 -> network_runner.py:98-153): the general front end behind the unchanged Python surface,
    including the `vectorizers` entry and `vectorize`."""
    import warnings
    from mycroft_precise_amd.network_runner import Listener
    from mycroft_precise_amd import vectorization as V
    from oracle import sonopy_restated as sr
    kw = dict(n_fft=1024, n_filt=40, n_mfcc=20)
    params = dict(window_t=0.1, hop_t=0.05, buffer_t=1.5, sample_rate=16000, sample_depth=2, vectorizer=2, use_delta=False, **kw)
    w = synth.make_weights(n_in=20, units=(20,), seed=13)
    saved = dict(P.pr.__dict__)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            path = _write_model(tmp_path, w, params, name='wide_front_end.npz')
            lis = Listener(path, 2048)
            assert (lis.pr.n_fft, lis.pr.n_filt, lis.pr.n_mfcc) == (1024, 40, 20)
            ref = ol.OracleListener(w, ol.Params(**kw))
            pcm = synth.stream_pcm(17, 36 * 1024)
            for u in range(36):
                chunk = pcm[u * 1024:(u + 1) * 1024].tobytes()
                assert abs(lis.update_raw(chunk) - ref.update_raw(chunk)) <= GUARD_RAW, u
            assert lis.mfccs.shape == (29, 20) and np.abs(lis.mfccs - ref.mfccs).max() <= TOL_FEAT32
            audio = pcm[:30000].astype(np.float64) / 32768.0
            want = sr.mfcc_spec(audio, 16000, (1600, 800), 1024, 40, 20)
            assert np.abs(V.vectorize_raw(audio) - want).max() <= 1e-9
            tail = sr.mfcc_spec(audio[-24000:], 16000, (1600, 800), 1024, 40, 20)      # vectorize keeps the last max_samples (vectorization.py:62-84)
            assert tail.shape == (29, 20) and np.abs(V.vectorize(audio) - tail).max() <= 1e-9
    finally:
        P.pr.__dict__.clear(); P.pr.__dict__.update(saved)


def test_speechpy_batched_streams_match_oracle(stock_weights):
    """The same front end for B lock-step streams (fused and two-launch, pe_update_many), vs the oracle."""
    from mycroft_precise_amd.network_runner import BatchedListener
    snap = P.pr.copy()
    snap.__dict__['vectorizer'] = P.Vectorizer.speechpy_mfccs
    kinds = (['tone_noise'] * 30) + ['zeros', 'square', 'square', 'quiet', 'quiet', 'zeros', 'tone_noise']
    n_up = 40
    pcm = _stream_batch(kinds, n_up)
    hip = BatchedListener(stock_weights, len(kinds), params=snap)
    two = BatchedListener(stock_weights, len(kinds), params=snap)
    two.engine.set_fused(False)
    many = BatchedListener(stock_weights, len(kinds), params=snap)
    many.engine.reserve_updates(8, 1024)
    ref = ol.BatchedOracle(stock_weights, len(kinds), ol.Params(vectorizer=3))
    outs = []
    for u in range(n_up):
        raw = hip.update_raw(pcm[u])
        want = ref.update_raw(pcm[u])
        assert np.abs(raw.astype(np.float64) - want).max() <= GUARD_RAW, u
        assert np.array_equal(raw, two.update_raw(pcm[u])), u
        assert np.abs(hip.engine.get_vectors().astype(np.float64) - ref.mfccs).max() <= TOL_FEAT32, u
        outs.append(raw)
    for u in range(0, n_up, 8):
        assert np.array_equal(many.engine.update_many(pcm[u:u + 8]), np.stack(outs[u:u + 8])), u
    q, kc, ke = hip.engine.stream_state()
    assert np.all(q + 800 * (kc - ke).astype(np.int64) == ref.window_audio.shape[1])


# ---- arbitrary float samples into Listener.update (network_runner.py:126-127) --------------------------------------
def test_listener_accepts_any_float_ndarray_audio(model_file):
    """load_audio-style k/32767 samples and a mixed / rescaled float64 signal, fed as ndarrays the way
    scripts/train_incremental.py does; fixture = the reference's own Listener on the same arrays."""
    from mycroft_precise_amd.network_runner import Listener
    g = golden('listener_float_audio.npz')
    for name in ('div32767', 'mixed64'):
        audio = g['audio_' + name]
        lis = Listener(model_file, 2048)
        raws = [lis.update_raw(audio[off:off + 1024]) for off in range(0, len(audio) - 1023, 1024)]
        assert np.abs(np.array(raws) - g['raw_' + name]).max() <= GUARD_RAW, name
        assert np.abs(lis.mfccs - g['ring_last_' + name]).max() <= 1e-9, name      # float64 path end to end
        assert len(lis.window_audio) == int(g['leftover_' + name])
    # a stream that starts as PCM bytes (device-resident state) and continues with float samples
    lis = Listener(model_file, 2048)
    pcm, audio = g['pcm'], g['pcm'].astype(np.float32) / np.float32(32767.0)
    raws = []
    for u in range(30):
        raws.append(lis.update_raw(pcm[u * 1024:(u + 1) * 1024].tobytes() if u < 12 else audio[u * 1024:(u + 1) * 1024]))
    assert np.abs(np.array(raws) - g['raw_bytes_then_float']).max() <= GUARD_RAW
    # exact k/32768 floats (buffer_to_audio output) stay on the int16 device path; an empty ndarray appends nothing
    lis2 = Listener(model_file, 2048)
    a = lis2.update_raw(pcm[:1024].astype(np.float32) / np.float32(32768.0))
    lis3 = Listener(model_file, 2048)
    assert a == lis3.update_raw(pcm[:1024].tobytes()) and not lis2._float_mode
    assert lis3.update_raw(np.array([], dtype=np.float32)) == a


def test_listener_runner_can_be_replaced_after_construction(model_file, stock_weights):
    """scripts/train_incremental.py:87-88 assigns ``listener.runner`` on a live Listener: predictions must come
    from the new runner, the stream state (leftover audio + feature window) stays."""
    from mycroft_precise_amd.network_runner import Listener, HipRunner
    other = synth.make_weights(seed=77)
    pcm = synth.stream_pcm(3, 40 * 1024)
    a = Listener(model_file, 2048)                                     # stock weights, then swapped at update 17
    want_stock, want_other = ol.OracleListener(stock_weights), ol.OracleListener(other)
    for u in range(40):
        chunk = pcm[u * 1024:(u + 1) * 1024].tobytes()
        if u == 17:
            a.runner = HipRunner(weights=other)
        if u == 29:
            a.runner = keras_gru.make_runner_cls(stock_weights)('x')   # a foreign runner: MFCC stays on the GPU
        got = a.update_raw(chunk)
        ws, wo = want_stock.update_raw(chunk), want_other.update_raw(chunk)
        assert abs(got - (wo if 17 <= u < 29 else ws)) <= GUARD_RAW, u
    assert np.abs(a.mfccs - want_stock.mfccs).max() <= TOL_FEAT32


def test_listener_mfccs_can_be_assigned(model_file, stock_weights):
    """``listener.mfccs = window`` (a plain attribute in the reference, network_runner.py:104): the device's window
    follows, the leftover samples stay, and the next updates agree with an oracle listener treated the same way."""
    from mycroft_precise_amd.network_runner import Listener
    pcm = synth.stream_pcm(8, 30 * 1024)
    a, ref = Listener(model_file, 2048), ol.OracleListener(stock_weights)
    window = np.random.default_rng(6).normal(0, 3, (29, 13)).astype(np.float32).astype(np.float64)
    for u in range(30):
        chunk = pcm[u * 1024:(u + 1) * 1024].tobytes()
        if u == 11:
            a.mfccs = window
            ref.mfccs = window.copy()
            assert len(a.window_audio) == len(ref.window_audio)
        assert abs(a.update_raw(chunk) - ref.update_raw(chunk)) <= GUARD_RAW, u
    assert np.abs(a.mfccs - ref.mfccs).max() <= TOL_FEAT32


# ---- full-size properties (BASELINE configs[1]: 4096 streams on one GPU) ----------------------------
def test_full_batch_4096_streams_properties(stock_weights):
    from mycroft_precise_amd.network_runner import BatchedListener
    B, n_up, n_check = 4096, 34, 48
    rng = np.random.default_rng(2024)
    base = synth.batch_pcm(n_check, n_up)                       # [n_up, n_check, 1024]
    # every stream is a copy of one of the n_check seeded streams, in a shuffled order
    owner = rng.integers(0, n_check, B)
    owner[:n_check] = np.arange(n_check)
    hip = BatchedListener(stock_weights, B)
    ref = ol.BatchedOracle(stock_weights, n_check)
    first_pass = []
    for u in range(n_up):
        raw = hip.update_raw(base[u][owner])
        want = ref.update_raw(base[u])
        assert raw.shape == (B,) and np.all(np.isfinite(raw))
        # (a) parity with the oracle on the seeded streams
        assert np.abs(raw[:n_check] - want).max() <= GUARD_RAW, u
        # (b) streams are independent: identical input -> bit-identical output wherever it sits
        assert np.array_equal(raw, raw[:n_check][owner]), u
        first_pass.append(raw)
    # (c) clear() + same input -> bit-identical outputs (stateless apart from the stream state)
    hip.clear()
    for u in range(n_up):
        assert np.array_equal(hip.update_raw(base[u][owner]), first_pass[u]), u
    info = hip.engine.info()
    assert info.n_streams == B and info.ring_slots == 32 and info.device_bytes > B * 2048


def _full_size_run(weights, B, n_up, n_check, tol, guard_equal_positions=True, oracle_params=None, gru_tiling=-1, **listener_kw):
    """Size-independent properties at a BASELINE batch size: every stream is a copy of one of n_check seeded
    streams (shuffled), so (a) the seeded ones are checked against the oracle, (b) identical input must give
    bit-identical output wherever it sits in the batch, (c) clear + replay is bit-identical."""
    from mycroft_precise_amd.network_runner import BatchedListener
    rng = np.random.default_rng(B + n_check)
    base = synth.batch_pcm(n_check, n_up)
    owner = rng.integers(0, n_check, B)
    owner[:n_check] = np.arange(n_check)
    hip = BatchedListener(weights, B, **listener_kw)
    if gru_tiling >= 0:
        hip.engine.set_gru_tiling(gru_tiling)
    ref = ol.BatchedOracle(weights, n_check, oracle_params)
    first, worst = [], 0.0
    for u in range(n_up):
        raw = hip.update_raw(base[u][owner])
        want = ref.update_raw(base[u])
        assert raw.shape == (B,) and np.all(np.isfinite(raw))
        worst = max(worst, float(np.abs(raw[:n_check] - want).max()))
        assert np.array_equal(raw, raw[:n_check][owner]), u
        first.append(raw)
    assert worst <= tol, worst
    hip.clear()
    for u in range(n_up):
        assert np.array_equal(hip.update_raw(base[u][owner]), first[u]), u
    return hip, base, owner, first


# ---- pe_set_gru_tiling(e, 2): the float32 network on the bf16 matrix pipe (gru_x3_device.h) -------------------------
def test_x3_network_matches_oracle_at_the_float32_guard(stock_weights):
    """Every operand as three bf16 pieces (hi + mid + lo == the float32 value exactly), six piece products per
    multiplication, float32 accumulate / gates / state: the float32 kernels' guard against the float32 oracle, and a
    distance to a FLOAT64 evaluation of the same windows that is the float32 kernels' (not a reduced-precision mode)."""
    from mycroft_precise_amd._lib import HipEngine
    kinds = (['tone_noise'] * 30) + ['zeros', 'square', 'square', 'quiet', 'quiet', 'zeros', 'tone_noise']
    n_up = 40
    pcm = _stream_batch(kinds, n_up)
    x3 = HipEngine(P.pr, stock_weights, n_streams=len(kinds))
    x3.set_gru_tiling(2)
    f32 = HipEngine(P.pr, stock_weights, n_streams=len(kinds))
    assert x3.gru_tiling() == 2 and f32.gru_tiling() == 1
    ref = ol.BatchedOracle(stock_weights, len(kinds))
    d64 = {'x3': 0.0, 'f32': 0.0}
    for u in range(n_up):
        px, pf = x3.update(pcm[u]), f32.update(pcm[u])
        want = ref.update_raw(pcm[u])
        assert np.abs(px.astype(np.float64) - want).max() <= GUARD_RAW, u
        feats = x3.get_vectors()
        assert np.array_equal(feats, f32.get_vectors())
        p64 = keras_gru.predict(feats, stock_weights, dtype=np.float64)[:, 0]
        d64['x3'] = max(d64['x3'], float(np.abs(px - p64).max()))
        d64['f32'] = max(d64['f32'], float(np.abs(pf - p64).max()))
    print('max |p - p_float64|: x3 %.3g, f32 MFMA %.3g' % (d64['x3'], d64['f32']))
    assert d64['x3'] <= 2 * d64['f32'] + 2e-7
    x3.close(); f32.close()


def test_x3_every_entry_point_agrees_bitwise(stock_weights):
    """One arithmetic for an engine's launches: pe_update == pe_update_many == pe_predict on the held windows ==
    pe_evaluate's strided windows (kRing / kFeats / kRows modes of gru_tile_x3), ragged last tile included."""
    from mycroft_precise_amd._lib import HipEngine
    n, depth, chunk = 70, 6, 1024
    kinds = ['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet']
    pcm = _stream_batch(kinds, 36, chunk)
    a = HipEngine(P.pr, stock_weights, n_streams=n)
    b = HipEngine(P.pr, stock_weights, n_streams=n)
    a.set_gru_tiling(2); b.set_gru_tiling(2)
    b.reserve_updates(depth, chunk)
    for u in range(0, 36, depth):
        want = np.stack([a.update(pcm[u + i]) for i in range(depth)])
        assert np.array_equal(b.update_many(pcm[u:u + depth]), want), u
        assert np.array_equal(a.predict(a.get_vectors())[:, 0], want[-1]), u
    a.close(); b.close()
    # offline evaluation: windows of one row sequence == the same windows as an explicit batch
    from oracle import sonopy_restated as so
    eng = HipEngine(P.pr, stock_weights, n_streams=1)
    eng.set_gru_tiling(2)
    audio = synth.stream_pcm(21, 16000 * 5).astype(np.float32) / np.float32(32768.0)
    got = eng.evaluate(audio, 2)
    mf = so.mfcc_spec(audio.astype(np.float64), 16000, (1600, 800), num_filt=20, fft_size=512, num_coeffs=13)
    windows = np.stack([mf[i - 29:i] for i in range(29, len(mf), 2)]).astype(np.float32)
    assert got.shape == (len(windows), 1)
    assert np.abs(got - keras_gru.predict(windows, stock_weights)).max() <= GUARD_RAW
    eng.close()


@pytest.mark.parametrize('units,n_in', [(1, 13), (7, 5), (16, 13), (17, 15), (20, 13), (20, 1)])
def test_x3_widths_and_input_sizes(units, n_in):
    """Units 1..20 (three full tiles + the quarter tile, zero-padded) and 1..15 inputs (the bias rides as pseudo-feature
    n_in); explicit batches of ragged size against the oracle."""
    from mycroft_precise_amd._lib import HipEngine
    w = synth.make_weights(n_in=n_in, units=(units,), seed=300 + units + n_in)
    hpr = P.pr.copy()
    hpr.__dict__.update(n_mfcc=n_in)
    eng = HipEngine(hpr, w, n_streams=1)
    eng.set_gru_tiling(2)
    rng = np.random.default_rng(units)
    for n in (1, 17, 50):
        x = rng.normal(0, 2, (n, 29, n_in)).astype(np.float32)
        x[:, :, 0] -= 20.0                      # (the log-energy coefficient is large)
        assert np.abs(eng.predict(x) - keras_gru.predict(x, w)).max() <= GUARD_RAW
    eng.close()


def test_x3_edge_cases_saturation_huge_nonfinite_and_denormal(stock_weights):
    """The edge cases the f32-input kernels were already held to, in the XDL form (tiling 2, the automatic network above 16 384
    streams): saturation to EXACTLY 0.0 / 1.0 (ThresholdDecoder.decode short-circuits on those, threshold_decoder.py:46-47),
    features up to 1e30 (three bf16 pieces carry float32's exponent range), denormal and tiny weights (bf16 pieces of a float32
    denormal drop bits below 2^-133: absolute error <= 5e-41 per product), and non-finite features: the first piece of inf is
    inf, the remainder inf - inf = NaN, so a window that contains inf / NaN / |x| >= 3.39e38 comes out NaN -- never a finite
    wrong number -- and the other windows of the batch are untouched (the f32-input forms and numpy propagate inf / NaN their
    own way: no form promises more than "non-finite in, non-finite or saturated out")."""
    from mycroft_precise_amd._lib import HipEngine
    eng = HipEngine(P.pr, stock_weights, n_streams=1)
    eng.set_gru_tiling(2)
    f32 = HipEngine(P.pr, stock_weights, n_streams=1)
    for scale in (1e4, 1e30):
        big = np.full((6, 29, 13), scale, np.float32)
        big[2:4] *= -1
        big[4:, :, ::2] *= -1
        want = keras_gru.predict(big, stock_weights)
        assert np.array_equal(eng.predict(big), want), scale
        assert np.array_equal(f32.predict(big), want), scale
        assert set(np.unique(want)) <= {0.0, 1.0}
    rng = np.random.default_rng(77)
    x = rng.normal(0, 3, (40, 29, 13)).astype(np.float32)
    clean = eng.predict(x)
    bad = x.copy()
    bad[3, 5, 2] = np.inf
    bad[11, 0, 12] = -np.inf
    bad[17, 28, 0] = np.nan
    bad[22, 9, 7] = np.float32(3.4e38)
    got = eng.predict(bad)
    hit = np.zeros(40, bool)
    hit[[3, 11, 17, 22]] = True
    assert np.all(np.isnan(got[hit]))
    assert np.array_equal(got[~hit], clean[~hit])
    # denormal / tiny weights
    w = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in stock_weights.items()}
    k0, r0, b0 = [a.copy() for a in w['gru'][0]]
    k0[::3, ::5] = np.float32(1e-40); k0[1::3, 1::5] = np.float32(-3e-39); r0[::4, ::7] = np.float32(7e-41); r0[2::4, 3::7] = np.float32(1.2e-38)
    b0[::6] = np.float32(2e-42)
    w['gru'] = [(k0, r0, b0)]
    den = HipEngine(P.pr, w, n_streams=1)
    den.set_gru_tiling(2)
    assert np.abs(den.predict(x) - keras_gru.predict(x, w)).max() <= GUARD_RAW
    den.close(); eng.close(); f32.close()


def test_decoder_bin_flips_between_network_forms(model_file, stock_weights):
    """ThresholdDecoder.decode is a step function of logit(raw) (6400 bins, threshold_decoder.py:45-57) and the three forms of
    the float32 network agree to float32 summation order: how many decoded values differ AT ALL when an engine crosses a size
    at which the automatic form changes (8192 / 16 384 streams)?  Counted on the reference fixture's streams (280 decodes, vs
    the reference's own decoded values) and on 256 synthetic streams x 48 updates (12 288 decodes, forms against each other);
    never more than one bin.  The counts are recorded in INTEGRATION.md section 7."""
    from mycroft_precise_amd._lib import HipEngine
    from mycroft_precise_amd.threshold_decoder import ThresholdDecoder
    g = golden('listener_chunk2048.npz')
    dec = ThresholdDecoder(P.pr.threshold_config, P.pr.threshold_center)
    n = len(g['streams'])
    pcm = g['pcm'].reshape(n, -1, 1024)
    fixture = {}
    for tiling in (0, 1, 2):
        eng = HipEngine(P.pr, stock_weights, n_streams=n)
        eng.set_gru_tiling(tiling)
        decs = np.array([[dec.decode(r) for r in eng.update(np.ascontiguousarray(pcm[:, u]))] for u in range(pcm.shape[1])]).T
        fixture[tiling] = sum(_decode_flips(decs[i], g['decoded'][i]) for i in range(n))
        eng.close()
    B, n_up = 256, 48
    base = synth.batch_pcm(B, n_up)
    decoded = {}
    for tiling in (0, 1, 2):
        eng = HipEngine(P.pr, stock_weights, n_streams=B)
        eng.set_gru_tiling(tiling)
        eng.set_decoder(dec)
        decoded[tiling] = np.stack([eng.decode(eng.update(base[u])) for u in range(n_up)])
        eng.close()
    pairs = {(a, b): _decode_flips(decoded[a], decoded[b]) for a, b in ((0, 1), (1, 2), (0, 2))}
    print('decoder bin flips vs the reference fixture (280 decodes) by form:', fixture, '; between forms on %d decodes:' % (B * n_up), pairs)
    assert fixture[1] == DECODE_FLIPS['chunk2048']
    assert max(fixture.values()) <= 3 and max(pairs.values()) <= B * n_up // 100, (fixture, pairs)


def test_x3_refuses_what_it_has_no_kernel_for():
    from mycroft_precise_amd._lib import HipEngine
    for kw, w_kw in ((dict(), dict(units=(21,))), (dict(n_mfcc=16), dict(n_in=16)), (dict(use_delta=True), dict(n_in=26))):
        hpr = P.pr.copy()
        hpr.__dict__.update(kw)
        eng = HipEngine(hpr, synth.make_weights(seed=1, **w_kw), n_streams=4)
        with pytest.raises(NotImplementedError):
            eng.set_gru_tiling(2)
        assert eng.gru_tiling() in (0, 1)
        eng.close()
    eng = HipEngine(P.pr, synth.make_weights(seed=1), n_streams=4, gru_precision='bf16')
    with pytest.raises(NotImplementedError):
        eng.set_gru_tiling(2)
    assert eng.gru_tiling() == 1             # the stock bf16 network: five values per lane (gru_b20_device.h)
    eng.close()


def test_x3_is_the_automatic_form_of_engines_that_fill_the_machine(stock_weights):
    """With more stream tiles than the machine has SIMDs (above 16 384 streams on MI355X) the float32 network takes the XDL form
    by itself; smaller engines keep the f32-input MFMA kernels (their fused launch is worth more than the cheaper network)."""
    from mycroft_precise_amd._lib import HipEngine
    for n, want in ((4096, 1), (16384, 0), (20480, 2), (32768, 2), (65536, 2)):
        eng = HipEngine(P.pr, stock_weights, n_streams=n)
        assert eng.gru_tiling() == want, n
        eng.set_gru_tiling(0)
        assert eng.gru_tiling() == 0
        eng.close()


@pytest.mark.parametrize('tiling', [-1, 0])
def test_capacity_batch_65536_streams_properties(stock_weights, tiling):
    """The capacity point of the metric (max concurrent real-time streams; bench.py extra_configs): 65536 streams, float64
    front end + float32 network, one wave per tile -- the automatic form (float32 products on the bf16 pipe, two launches)
    and the classic tiling on the f32-input MFMAs (fused launch): the size-independent properties, and pe_set_fused(0)
    == the default sequencing at this size."""
    from mycroft_precise_amd._lib import HipEngine
    hip, base, owner, first = _full_size_run(stock_weights, 65536, 33, 24, GUARD_RAW, gru_tiling=tiling)
    assert hip.engine.gru_tiling() == (2 if tiling < 0 else 0)
    two = HipEngine(P.pr, stock_weights, n_streams=65536)
    two.set_gru_tiling(tiling)
    two.set_fused(False)
    for u in range(33):
        assert np.array_equal(two.update(base[u][owner]), first[u]), u
    two.close()


def test_full_batch_float32_front_end_4096_streams_properties(stock_weights):
    """mfcc_precision='f32' beside the float32 network in the fused launch at the BASELINE batch size: the size-independent
    properties (identical input => bit-identical output at any position, clear + replay) -- the float32 frame role is the
    one whose results turned out to depend on what runs beside it (profiles/round4/r4v_b20_fused_corruption.log)."""
    _full_size_run(stock_weights, 4096, 33, 16, TOL_RAW, mfcc_precision='f32')


def test_full_batch_general_front_end_4096_streams_properties():
    """A non-stock .params file at the BASELINE batch size (params.py:28-118: n_fft 1024, 40 filters, 20 coefficients --
    the general front end, 32-float feature rows, the four-wave network over them): the size-independent properties and
    the oracle on the seeded streams; then the one-wave network shape on the same updates, bit for bit."""
    import warnings
    from mycroft_precise_amd._lib import HipEngine
    kw = dict(n_fft=1024, n_filt=40, n_mfcc=20)
    hpr = P.pr.copy()
    hpr.__dict__.update(kw)
    w = synth.make_weights(n_in=20, units=(20,), seed=11)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        hip, base, owner, first = _full_size_run(w, 4096, 33, 16, GUARD_RAW, oracle_params=ol.Params(**kw), params=hpr)
        one = HipEngine(hpr, w, n_streams=4096)
    one.set_gru_waves(1)
    for u in range(33):
        assert np.array_equal(one.update(base[u][owner]), first[u]), u
    one.close()


@pytest.mark.parametrize('tiling', [0, 2], ids=['f32mfma', 'xdl'])
def test_full_batch_wide_gru_4096_streams_properties(tiling):
    """BASELINE configs[3]: 256 x 2 layers at 4096 streams = 256 workgroups sharing one L2-resident weight
    stream -- where a stale read or an ordering slip between workgroups would show."""
    w = synth.make_weights(units=(256, 256), seed=5)
    hip, base, owner, first = _full_size_run(w, 4096, 31, 16, GUARD_RAW, gru_tiling=tiling)
    assert hip.engine.gru_tiling() == tiling
    # Runner.predict on the windows the streams hold now == the streaming output, bit for bit
    feats = hip.engine.get_vectors()
    assert np.array_equal(hip.engine.predict(feats)[:, 0], first[-1])


@pytest.mark.parametrize('B,ring', [(8192, 'f32'), (8192, 'bf16'), (65536, 'bf16')])
def test_full_batch_bf16_properties(stock_weights, B, ring):
    """BASELINE configs[4] per-GPU shapes: 8192 and 65536 streams, bf16 network + float32 front end, feature rows
    kept as float32 or as bf16 (32 bytes per frame and stream); the launch policy differs between the two sizes
    (tiles <= / > compute units)."""
    hip, base, owner, first = _full_size_run(stock_weights, B, 32, 24, TOL_BF16, mfcc_precision='f32', gru_precision='bf16',
                                             ring_precision=ring)
    # both launch shapes of the network agree bit for bit at this size too
    hip.engine.set_fused(False)
    hip.clear()
    for u in range(8):
        assert np.array_equal(hip.update_raw(base[u][owner]), first[u]), u


def test_full_batch_update_many_4096_streams_x8(stock_weights):
    """pe_update_many at the bench's shape (4096 streams x 8 updates per call) == single updates, bit for bit."""
    from mycroft_precise_amd._lib import HipEngine
    B, depth, n_check = 4096, 8, 32
    rng = np.random.default_rng(99)
    base = synth.batch_pcm(n_check, 5 * depth)
    owner = rng.integers(0, n_check, B)
    owner[:n_check] = np.arange(n_check)
    a = HipEngine(P.pr, stock_weights, n_streams=B)
    b = HipEngine(P.pr, stock_weights, n_streams=B)
    b.reserve_updates(depth, 1024)
    # 256 tiles x 8 updates = 2048 windows per call > 4 per compute unit: the reserved engine takes the float32 network on the
    # bf16 pipe for all of its launches (round 6); bit identity holds within a form, so the single-update engine is put on it too
    assert b.gru_tiling() == 2 and a.gru_tiling() == 1
    a.set_gru_tiling(2)
    ref = ol.BatchedOracle(stock_weights, n_check)
    for r in range(5):
        pcm = np.ascontiguousarray(base[r * depth:(r + 1) * depth][:, owner])
        want = np.stack([a.update(pcm[i]) for i in range(depth)])
        got = b.update_many(pcm)
        assert np.array_equal(got, want), r
        for i in range(depth):
            assert np.abs(got[i][:n_check] - ref.update_raw(base[r * depth + i])).max() <= GUARD_RAW
    for x, y in zip(a.stream_state(), b.stream_state()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_bench_two_ranks_on_one_gpu_shards_streams_correctly():
    """bench.py --gpus 2 under torch.distributed.run with both ranks on cuda:0 (PE_BENCH_SHARED_GPU=1, gloo): the
    N > 1 control flow with the REAL engine -- global stream count, rank-ordered gather, and rank 1's
    probabilities equal to a single-rank run over streams 512..1023."""
    import json
    env = dict(os.environ, PE_BENCH_SHARED_GPU='1', PE_BENCH_DUMP=os.path.join(REPO, 'gpurun_out', 'bench2_probs.npy'),
               PYTHONPATH=REPO + os.pathsep + os.environ.get('PYTHONPATH', ''))
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    common = ['--steps', '6', '--warmup', '30', '--streams', '512', '--no-cpu-baseline']
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', '29517', os.path.join(REPO, 'bench.py'),
                          '--gpus', '2'] + common, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['global_streams'] == 1024 and line['scaling'] == 'weak'
    assert line['outputs_finite'] and line['value'] > 0
    both = np.load(env['PE_BENCH_DUMP'])                     # [steps, 1024] gathered on rank 0
    assert both.shape == (6, 1024)
    env1 = dict(env, PE_BENCH_DUMP=os.path.join(REPO, 'gpurun_out', 'bench1_probs.npy'), PE_BENCH_FIRST_STREAM='512')
    one = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1'] + common, env=env1, cwd=REPO,
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    solo = np.load(env1['PE_BENCH_DUMP'])
    assert solo.shape == (6, 512) and np.array_equal(both[:, 512:], solo)
    assert not np.array_equal(both[:, :512], both[:, 512:])          # the two shards really are different streams


# ---- streams that advance independently (pe_update_subset) ------------------------------------------------------------------------
def _cadence_run(hip, refs, rng, n_calls, sizes, audio, pos, check_every_raw=True):
    """`n_calls` subset calls: each picks one chunk length and a random set of streams; every active stream is compared with
    its own OracleListener (network_runner.py:125-153 restated), inactive streams must not move."""
    n = len(refs)
    worst = 0.0
    for call in range(n_calls):
        chunk = int(sizes[int(rng.integers(0, len(sizes)))])
        p_active = rng.random(n)                                   # per-stream cadence: some streams speak often, some rarely
        active = np.nonzero(rng.random(n) < hip._cadence * (0.2 + p_active))[0]
        active = active[(pos[active] + chunk) <= audio.shape[1]]
        rng.shuffle(active)                                        # (the order of the ids is the order of rows and outputs)
        if active.size == 0:
            assert hip.update_raw(np.empty((0, chunk), np.int16), streams=[]).size == 0
            continue
        pcm = np.stack([audio[s, pos[s]:pos[s] + chunk] for s in active])
        before = hip.engine.stream_state()
        got = hip.update_raw(pcm, streams=active)
        after = hip.engine.stream_state()
        idle = np.setdiff1d(np.arange(n), active)
        for b, a in zip(before, after):
            assert np.array_equal(b[idle], a[idle]), call          # streams without audio did not move
        for k, s in enumerate(active):
            want = refs[s].update_raw(pcm[k].tobytes())
            worst = max(worst, abs(float(got[k]) - want))
            pos[s] += chunk
        assert worst <= GUARD_RAW, (call, chunk, worst)
    return worst


@pytest.mark.parametrize('n,cadence,sizes', [(64, 0.5, (1024,)), (64, 0.6, (1024, 2048, 640, 801, 160)), (200, 0.3, (1024, 512))])
def test_streams_advance_independently(stock_weights, n, cadence, sizes):
    """n streams fed at n different random cadences and chunk phases for 200 calls, every stream against its own
    OracleListener at GUARD_RAW; leftover length, counters and feature windows of every stream at the end."""
    from mycroft_precise_amd.network_runner import BatchedListener
    rng = np.random.default_rng(20260930 + n + len(sizes))
    kinds = (['tone_noise'] * (n - 4)) + ['zeros', 'square', 'quiet', 'tone_noise']
    audio = np.stack([synth.stream_pcm(s, 200 * 1024, k) for s, k in enumerate(kinds)])
    hip = BatchedListener(stock_weights, n)
    hip._cadence = cadence
    refs = [ol.OracleListener(stock_weights) for _ in range(n)]
    pos = np.zeros(n, dtype=np.int64)
    # phases: every stream starts with a private odd-sized first chunk (one call per length)
    first = rng.integers(1, 1500, n)
    for length in np.unique(first):
        ids = np.nonzero(first == length)[0]
        pcm = np.stack([audio[s, :length] for s in ids])
        got = hip.update_raw(pcm, streams=ids)
        for k, s in enumerate(ids):
            assert abs(float(got[k]) - refs[s].update_raw(pcm[k].tobytes())) <= GUARD_RAW
            pos[s] = length
    _cadence_run(hip, refs, rng, 200, sizes, audio, pos)
    q, kc, ke = hip.engine.stream_state()
    left = np.array([len(r.window_audio) for r in refs])
    assert np.array_equal(q + 800 * (kc - ke).astype(np.int64), left)
    want = np.stack([r.mfccs for r in refs])
    assert np.abs(hip.engine.get_vectors().astype(np.float64) - want).max() <= TOL_FEAT32
    assert len(np.unique(pos)) > n // 2                            # the streams really are at different places
    hip.engine.close()


def test_subset_of_all_streams_equals_full_update_bitwise(stock_weights):
    """pe_update_subset over every stream in order == pe_update, bit for bit, fused and in two launches, interleaved with full
    updates and pe_update_many on the same engine; a permutation of the ids permutes rows and outputs; the host entry point
    refuses ids out of range / named twice, and an empty chunk is EOF."""
    from mycroft_precise_amd._lib import HipEngine
    n, n_up = 70, 36
    pcm = _stream_batch(['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet'], n_up)
    a = HipEngine(P.pr, stock_weights, n_streams=n)
    b = HipEngine(P.pr, stock_weights, n_streams=n)
    c = HipEngine(P.pr, stock_weights, n_streams=n)
    c.set_fused(False)
    perm = np.random.default_rng(3).permutation(n)
    ids = np.arange(n)
    for u in range(n_up):
        want = a.update(pcm[u])
        if u % 3 == 0:
            got = b.update(pcm[u])                                 # a full update between subset calls
        elif u % 3 == 1:
            got = b.update_subset(ids, pcm[u])
        else:
            got = np.empty(n, np.float32)
            got[perm] = b.update_subset(perm, pcm[u][perm])
        assert np.array_equal(got, want), u
        two = np.empty(n, np.float32)
        half = perm[:n // 2], perm[n // 2:]                        # the same update as two calls over disjoint halves
        for h in half:
            two[h] = c.update_subset(h, pcm[u][h])
        assert np.array_equal(two, want), u
    for x, y, z in zip(a.stream_state(), b.stream_state(), c.stream_state()):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    assert np.array_equal(a.get_vectors(), b.get_vectors()) and np.array_equal(a.get_vectors(), c.get_vectors())
    with pytest.raises(ValueError):
        b.update_subset([0, n], pcm[0][:2])
    with pytest.raises(ValueError):
        b.update_subset([3, 3], pcm[0][:2])
    with pytest.raises(EOFError):
        b.update_subset([1], np.empty((1, 0), np.int16))
    assert b.update_subset([], np.empty((0, 1024), np.int16)).size == 0
    for x, y in zip(a.stream_state(), b.stream_state()):           # the refused calls moved nothing
        assert np.array_equal(x, y)
    a.close(); b.close(); c.close()


@pytest.mark.parametrize('kw', [dict(gru_precision='bf16', ring_precision='bf16', mfcc_precision='f32'), dict(units=(256, 256)),
                                dict(params=dict(n_fft=1024, n_filt=40, n_mfcc=20)), dict(use_delta=True), dict(tiling=2)],
                         ids=['bf16', 'wide256x2', 'general_fft1024', 'use_delta', 'x3'])
def test_subset_updates_in_every_network_and_front_end(kw):
    """The id indirection sits in every network prologue and both front ends: bf16 rows + five-values network, the wide streamed
    network, the general front end (32-float rows), delta inputs, the float32 network on the bf16 pipe -- each: random subsets at
    random chunk lengths; streams 0..5 also run alone on private single-stream engines, which must see the same bits (one stream
    alone vs the same stream at some position of some tile: same arithmetic per stream)."""
    import warnings
    from mycroft_precise_amd._lib import HipEngine
    kw = dict(kw)
    hpr = P.pr.copy()
    hpr.__dict__.update(kw.pop('params', {}))
    if kw.pop('use_delta', False):
        hpr.__dict__['use_delta'] = True
    tiling = kw.pop('tiling', None)
    units = kw.pop('units', (20,))
    n_in = hpr.n_mfcc * (2 if hpr.use_delta else 1)
    w = synth.make_weights(n_in=n_in, units=units, seed=13)
    n = 40
    rng = np.random.default_rng(77)
    audio = np.stack([synth.stream_pcm(s, 60 * 1024) for s in range(n)])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        eng = HipEngine(hpr, w, n_streams=n, **kw)
        solo = [HipEngine(hpr, w, n_streams=1, **kw) for _ in range(6)]        # private single-stream engines for streams 0..5
    if tiling is not None:
        eng.set_gru_tiling(tiling)
        for e1 in solo:
            e1.set_gru_tiling(tiling)
    pos = np.zeros(n, dtype=np.int64)
    tol = TOL_BF16 if kw.get('gru_precision') == 'bf16' else 0.0
    for call in range(60):
        chunk = int(rng.choice([1024, 1024, 2048, 800]))
        active = np.nonzero(rng.random(n) < 0.5)[0]
        active = active[pos[active] + chunk <= audio.shape[1]]
        if active.size == 0:
            continue
        rng.shuffle(active)
        pcm = np.stack([audio[s, pos[s]:pos[s] + chunk] for s in active])
        got = eng.update_subset(active, pcm)
        assert np.isfinite(got).all()
        for k, s in enumerate(active):
            if s < len(solo):
                want = solo[s].update(pcm[k][None])[0]
                # one stream alone vs the same stream inside a tile of sixteen: same arithmetic per stream, same bits
                assert abs(float(got[k]) - float(want)) <= tol, (call, s)
            pos[s] += chunk
    for s, e1 in enumerate(solo):
        assert np.array_equal(e1.get_vectors()[0], eng.get_vectors()[s])
        e1.close()
    eng.close()


def test_call_numbers_renumbered_before_they_wrap(stock_weights):
    """Records carry the number of the call that wrote them; the count is renumbered in place at a threshold (default 0x7fff0000;
    here 40, crossed several times) -- with streams that were idle across the renumbering."""
    from mycroft_precise_amd._lib import HipEngine
    n, n_up = 37, 120
    pcm = _stream_batch(['tone_noise'] * n, n_up)
    a = HipEngine(P.pr, stock_weights, n_streams=n)
    b = HipEngine(P.pr, stock_weights, n_streams=n)
    b.set_renumber_at(40)
    odd, even = np.arange(1, n, 2), np.arange(0, n, 2)
    for u in range(n_up):
        want = a.update(pcm[u])
        got = np.empty(n, np.float32)
        # even streams in one call, odd streams in a later one: between the two, half of the records are one call "older"
        got[even] = b.update_subset(even, pcm[u][even])
        got[odd] = b.update_subset(odd, pcm[u][odd])
        assert np.array_equal(got, want), u
    for x, y in zip(a.stream_state(), b.stream_state()):
        assert np.array_equal(x, y)
    with pytest.raises(ValueError):
        b.set_renumber_at(3)
    a.close(); b.close()


def test_subset_at_full_batch_costs_what_a_full_update_costs(stock_weights):
    """4096 of 4096 streams through pe_update_subset_device: bit-identical to pe_update_device, and within a few
    percent of its time -- the headline's fused launch with one extra load per stream."""
    import torch
    from mycroft_precise_amd._lib import HipEngine
    B, n_res = 4096, 16
    dev = torch.device('cuda', 0)
    base = synth.batch_pcm(64, n_res)                                           # [n_res][64][1024]
    pcm = torch.from_numpy(np.ascontiguousarray(np.tile(base, (1, B // 64, 1)))).to(dev)
    ids = torch.arange(B, dtype=torch.int32, device=dev)
    a = HipEngine(P.pr, stock_weights, n_streams=B)
    b = HipEngine(P.pr, stock_weights, n_streams=B)
    oa, ob = torch.zeros(B, device=dev), torch.zeros(B, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(eng, out, subset, n):
        for i in range(n):
            if subset:
                eng.update_subset_device(ids.data_ptr(), B, pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st)
            else:
                eng.update_device(pcm[i % n_res].data_ptr(), 1024, out.data_ptr(), st)

    for i in range(40):
        run(a, oa, False, 1); run(b, ob, True, 1)
        assert torch.equal(oa, ob), i
    times = {}
    for name, eng, out, subset in (('full', a, oa, False), ('subset', b, ob, True), ('full', a, oa, False), ('subset', b, ob, True)):
        run(eng, out, subset, 1500)                                            # (also takes the GPU out of idle)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(); run(eng, out, subset, 1500); ev1.record(); ev1.synchronize()
        times[name] = min(times.get(name, 1e9), ev0.elapsed_time(ev1) / 1500)
    print('full %.2f us, subset of all %.2f us per update' % (1e3 * times['full'], 1e3 * times['subset']))
    assert times['subset'] <= 1.06 * times['full'], times
    # a quarter of the streams active: the call costs less than the full one (cost follows the active streams)
    some = torch.arange(0, B, 4, dtype=torch.int32, device=dev)
    osome = torch.zeros(B // 4, device=dev)
    part = pcm[:, ::4].contiguous()
    for i in range(200):
        b.update_subset_device(some.data_ptr(), B // 4, part[i % n_res].data_ptr(), 1024, osome.data_ptr(), st)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(1000):
        b.update_subset_device(some.data_ptr(), B // 4, part[i % n_res].data_ptr(), 1024, osome.data_ptr(), st)
    ev1.record(); ev1.synchronize()
    assert ev0.elapsed_time(ev1) / 1000 <= 1.05 * times['full']
    a.close(); b.close()


@pytest.mark.parametrize('kw', [dict(), dict(mfcc_precision='f32'), dict(gru_precision='bf16', ring_precision='bf16', mfcc_precision='f32'),
                                dict(tiling=2), dict(units=(256,))],
                         ids=['stock_f64', 'front_end_f32', 'bf16', 'x3_two_launches', 'wide256'])
def test_kept_leftovers_equal_carried_leftovers(kw):
    """pe_update_device_keep: the leftover of Listener.update_vectors (network_runner.py:127-131) stays in the
    caller's chunk and the next call cuts its first frame's head from there.  Bit-identical to pe_update_device -- raw outputs
    after every call, feature windows and per-stream counters at the end -- through random chunk lengths (including ones that
    cannot hold a leftover: odd, 510 samples, and a tiny 160-sample chunk), with calls of every other style in between
    (plain device updates, subsets, a masked clear, get_vectors), each of which must find the kept leftovers."""
    import torch
    import warnings
    from mycroft_precise_amd._lib import HipEngine
    kw = dict(kw)
    tiling = kw.pop('tiling', None)
    units = kw.pop('units', (20,))
    w = synth.make_weights(n_in=P.pr.n_mfcc, units=units, seed=5)
    n, total = 70, 80 * 1024
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(2026)
    kinds = ['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet']
    audio = np.stack([synth.stream_pcm(s, total, k) for s, k in enumerate(kinds)])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        a = HipEngine(P.pr, w, n_streams=n, **kw)            # carried leftovers (pe_update_device)
        b = HipEngine(P.pr, w, n_streams=n, **kw)            # kept leftovers
    if tiling is not None:
        a.set_gru_tiling(tiling); b.set_gru_tiling(tiling)
    oa, ob = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    alive = []                                              # the promise: a kept chunk outlives the next call
    pos, n_keep = 0, 0
    for call in range(400):
        chunk = int(rng.choice([1024, 1024, 1024, 2048, 512, 600, 510, 1023, 160, 800]))
        if pos + chunk > total:
            break
        pcm = torch.from_numpy(np.ascontiguousarray(audio[:, pos:pos + chunk])).to(dev)
        alive = alive[-2:] + [pcm]
        style = rng.choice(['keep', 'keep', 'keep', 'plain', 'subset', 'clear', 'vectors'])
        a.update_device(pcm.data_ptr(), chunk, oa.data_ptr(), st)
        if style == 'keep':
            b.update_device(pcm.data_ptr(), chunk, ob.data_ptr(), st, keep=True); n_keep += 1
        elif style == 'subset':                              # every stream, shuffled: the same update through the id list
            ids = torch.from_numpy(rng.permutation(n).astype(np.int32)).to(dev)
            rows = pcm[ids.long()].contiguous()
            osub = torch.zeros(n, device=dev)
            b.update_subset_device(ids.data_ptr(), n, rows.data_ptr(), chunk, osub.data_ptr(), st)
            ob[ids.long()] = osub
            alive.append(rows)
        else:
            b.update_device(pcm.data_ptr(), chunk, ob.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.equal(oa, ob), (call, style, chunk)
        if style == 'clear':
            mask = (rng.random(n) < 0.3)
            a.clear(mask); b.clear(mask)
        elif style == 'vectors':
            assert np.array_equal(a.get_vectors(), b.get_vectors()), call
        pos += chunk
    assert n_keep > 20
    assert np.array_equal(a.get_vectors(), b.get_vectors())
    for x, y in zip(a.stream_state(), b.stream_state()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_kept_leftovers_at_full_batch(stock_weights):
    """4096 streams (the headline's fused launch) and 20 000 (two launches, the XDL network): keep calls over a ring of resident
    slabs, as bench.py cycles them, against plain calls -- same bits; then the engine is switched to update_many in the middle
    (leftovers move to the carry first)."""
    import torch
    from mycroft_precise_amd._lib import HipEngine
    dev = torch.device('cuda', 0)
    st = torch.cuda.current_stream().cuda_stream
    for B in (4096, 20000):
        n_res = 6
        base = synth.batch_pcm(64, n_res)
        reps = (B + 63) // 64
        pcm = torch.from_numpy(np.ascontiguousarray(np.tile(base, (1, reps, 1))[:, :B])).to(dev)
        a = HipEngine(P.pr, stock_weights, n_streams=B)
        b = HipEngine(P.pr, stock_weights, n_streams=B)
        oa, ob = torch.zeros(B, device=dev), torch.zeros(B, device=dev)
        for i in range(30):
            a.update_device(pcm[i % n_res].data_ptr(), 1024, oa.data_ptr(), st)
            b.update_device(pcm[i % n_res].data_ptr(), 1024, ob.data_ptr(), st, keep=True)
            assert torch.equal(oa, ob), (B, i)
        for x, y in zip(a.stream_state(), b.stream_state()):
            assert np.array_equal(x, y)
        assert np.array_equal(a.get_vectors(), b.get_vectors())
        a.close(); b.close()
    # keep calls, then several updates per call on the same (reserved) engine
    B = 256
    pcm = torch.from_numpy(np.ascontiguousarray(synth.batch_pcm(B, 12))).to(dev)        # [12][B][1024]
    a = HipEngine(P.pr, stock_weights, n_streams=B); a.reserve_updates(4, 1024)
    b = HipEngine(P.pr, stock_weights, n_streams=B); b.reserve_updates(4, 1024)
    oa, ob = torch.zeros(B, device=dev), torch.zeros(B, device=dev)
    ma, mb = torch.zeros(4, B, device=dev), torch.zeros(4, B, device=dev)
    for i in range(4):
        a.update_device(pcm[i].data_ptr(), 1024, oa.data_ptr(), st)
        b.update_device(pcm[i].data_ptr(), 1024, ob.data_ptr(), st, keep=True)
        assert torch.equal(oa, ob)
    # refilling ONE buffer breaks the promise: the second keep call on the same address is refused, nothing moves, and the
    # same chunk through the plain entry point continues the stream correctly
    with pytest.raises(ValueError):
        b.update_device(pcm[3].data_ptr(), 1024, ob.data_ptr(), st, keep=True)
    for x, y in zip(a.stream_state(), b.stream_state()):
        assert np.array_equal(x, y)
    a.update_many_device(pcm[4:8].data_ptr(), 1024, 4, ma.data_ptr(), st)
    b.update_many_device(pcm[4:8].data_ptr(), 1024, 4, mb.data_ptr(), st)
    assert torch.equal(ma, mb)
    for i in range(8, 12):
        a.update_device(pcm[i].data_ptr(), 1024, oa.data_ptr(), st)
        b.update_device(pcm[i].data_ptr(), 1024, ob.data_ptr(), st, keep=True)
        assert torch.equal(oa, ob)
    a.close(); b.close()


def test_rccl_calls_with_one_rank():
    """Every torch.distributed call of the N > 1 path, with bench.py's argument shapes, on the `nccl` backend (= RCCL) in a
    world of ONE rank on cuda:0 -- what a 1-GPU box can run of the RCCL path (zero executions so far).
    In a subprocess: the process group must not outlive the check."""
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get('PYTHONPATH', ''), MASTER_ADDR='127.0.0.1', MASTER_PORT='29577',
               RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tests', 'rccl_one_rank_check.py')], env=env, cwd=REPO,
                         capture_output=True, text=True, timeout=600)
    # (librccl prints its own path to stdout on the way out: any line, not the last one)
    assert out.returncode == 0 and any(l.startswith('ok: nccl backend (RCCL), 1 rank') for l in out.stdout.splitlines()), (out.stdout[-500:], out.stderr[-3000:])
    # ... and bench.py's own N > 1 plumbing over RCCL: PE_BENCH_FORCE_DIST=1 makes a world of one rank create its process group
    # (nccl, device_id) and issue every collective of the N > 1 path -- settle gather / all-gather, the timed region's gather,
    # the clock exchange, the asynchronous per-step gather
    import json
    env2 = dict(env, PE_BENCH_FORCE_DIST='1', MASTER_PORT='29578')
    b = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5', '--streams', '512',
                        '--no-cpu-baseline', '--gather-every-step', 'rccl'], env=env2, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stderr[-3000:]
    line = json.loads([l for l in b.stdout.splitlines() if l.startswith('{')][-1])
    assert line['collective_backend'] == 'nccl' and line['collective'] in ('gather', 'all_gather') and line['ranks_seen'] == 1
    assert line['parity']['ok'] and line['outputs_finite']
    ps = line['per_step_delivery']
    assert ps['mode'] == 'rccl' and ps['delivered_equals_device'] is True


def test_bench_starts_its_own_ranks_and_delivers_per_step():
    """`python bench.py --gpus 2` WITHOUT a launcher (the shape of the driver's N = 1 command): bench.py re-executes itself
    under torch.distributed.run on 127.0.0.1 with a free port.  Both ranks on cuda:0 over gloo (PE_BENCH_SHARED_GPU=1: this
    pool has one GPU per box).  One JSON line: two ranks seen, which collective carried the final gather, the headline's own
    parity object, global stream order (rank 1's shard == a solo run over streams 512..1023), and the second region of
    --gather-every-step delivering every step's probabilities unchanged."""
    import json
    dump = os.path.join(REPO, 'gpurun_out', 'bench2s_probs.npy')
    env = dict(os.environ, PE_BENCH_SHARED_GPU='1', PE_BENCH_DUMP=dump, PYTHONPATH=REPO + os.pathsep + os.environ.get('PYTHONPATH', ''))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    common = ['--steps', '20', '--warmup', '5', '--streams', '512', '--no-cpu-baseline']
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--gather-every-step', 'rccl'] + common,
                         env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['config']['global_streams'] == 1024
    assert line['collective_backend'] == 'gloo' and line['collective'] in ('gather', 'all_gather')
    assert line['streams_per_rank'] == [512, 512]
    assert line['parity']['ok'] and line['parity']['streams_checked'] == 256 and line['parity']['steps_checked'] == 20
    assert line['parity']['max_abs_err'] <= GUARD_RAW
    ps = line['per_step_delivery']
    assert ps['delivered_equals_device'] is True and ps['ms_per_step'] > 0
    both = np.load(dump)
    assert both.shape == (20, 1024)
    env1 = dict(env, PE_BENCH_DUMP=os.path.join(REPO, 'gpurun_out', 'bench1s_probs.npy'), PE_BENCH_FIRST_STREAM='512')
    one = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--gather-every-step', 'host'] + common, env=env1, cwd=REPO,
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    solo_line = json.loads([l for l in one.stdout.splitlines() if l.startswith('{')][-1])
    assert solo_line['parity']['ok'] and solo_line['per_step_delivery']['delivered_equals_device'] is True
    solo = np.load(env1['PE_BENCH_DUMP'])
    assert solo.shape == (20, 512) and np.array_equal(both[:, 512:], solo)
    # 'direct': the update's output pointer is a row of a pinned host ring -- the kernel's own final store is the delivery
    direct = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--gather-every-step', 'direct', '--no-extra-configs', '--no-batched'] + common,
                            env={k: v for k, v in env.items() if k != 'PE_BENCH_DUMP'}, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert direct.returncode == 0, direct.stderr[-3000:]
    dline = json.loads([l for l in direct.stdout.splitlines() if l.startswith('{')][-1])
    assert dline['per_step_delivery']['mode'] == 'direct' and dline['per_step_delivery']['delivered_equals_device'] is True
    # a rank count the node cannot serve is refused by name before anything is launched (no PE_BENCH_SHARED_GPU)
    env2 = {k: v for k, v in env.items() if k != 'PE_BENCH_SHARED_GPU'}
    import torch
    bad = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(torch.cuda.device_count() + 1)] + common, env=env2, cwd=REPO,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and 'exposes' in bad.stderr and 'Traceback' not in bad.stderr


# ---- the float32 frame role beside the bf16 matrix pipe: a gate, not a tool ----------------------------------------
def _position_soak(eng, n_updates, every=16):
    """Every stream gets the SAME audio, so every position must produce the same bits: raw outputs compared with stream 0
    on the device after every update, whole feature windows every `every` updates (a frame stays in the window for ~22)."""
    import torch
    dev = torch.device('cuda', 0)
    B = eng.n_streams
    n_res = 8
    base = synth.batch_pcm(1, n_res)
    pcm = torch.from_numpy(np.ascontiguousarray(base[:, 0, :])).to(dev)[:, None, :].expand(n_res, B, 1024).contiguous()
    out = torch.zeros(B, device=dev)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    bad_windows = 0
    for u in range(n_updates):
        eng.update_device(pcm[u % n_res].data_ptr(), 1024, out.data_ptr(), st)
        bad += (out != out[0]).sum()
        if (u + 1) % every == 0 or u == n_updates - 1:
            torch.cuda.synchronize()
            feats = eng.get_vectors()
            bad_windows += int(np.any(feats != feats[0], axis=(1, 2)).sum())
    torch.cuda.synchronize()
    assert np.isfinite(out.cpu().numpy()).all() and float(out[0]) > 0.0
    return int(bad.item()), bad_windows


@pytest.mark.parametrize('streams,updates', [(8192, 100), (65536, 16)])
def test_float32_frames_position_soak_beside_b20_network(stock_weights, streams, updates):
    """Round 4 found ~0.7 % of the float32 frames of a fused launch wrong beside the five-values bf16 network role when the
    frame role used packed float32 instructions; the cure (kernels.hip: PE_NO_PK_F32 on every kernel that hosts float32
    butterflies) has no isolated cause, so a compiler upgrade could bring it back silently.  This is tools/gpu_frame_soak.py
    as a gate: >= 2e6 float32 frames over the two sizes, five-values role on, bf16 rows, every position compared."""
    from mycroft_precise_amd._lib import HipEngine
    eng = HipEngine(P.pr, stock_weights, n_streams=streams, mfcc_precision='f32', gru_precision='bf16', ring_precision='bf16')
    assert eng.gru_tiling() == 1                        # five gate values per lane (gru_b20_device.h)
    bad_out, bad_win = _position_soak(eng, updates)
    eng.close()
    assert (bad_out, bad_win) == (0, 0), 'float32 frames disagree between positions: %d outputs, %d feature windows' % (bad_out, bad_win)


def test_general_float32_front_end_soak_beside_b20_network():
    """The general front end hosts float32 butterflies too (mfcc_general_stream_kernel<float>: compiled without packed float32
    since round 6) and may feed the five-values bf16 network; a SECOND engine's fused float32 + b20 launches run on another
    stream at the same time, so packed-free frame waves and b20 MFMA waves of different engines share compute units."""
    import threading
    import torch
    from mycroft_precise_amd._lib import HipEngine
    kw = dict(n_fft=256, n_filt=20, n_mfcc=13)
    hpr = P.pr.copy()
    hpr.__dict__.update(kw)
    w = synth.make_weights(n_in=13, units=(20,), seed=7)
    gen = HipEngine(hpr, w, n_streams=8192, mfcc_precision='f32', gru_precision='bf16', ring_precision='bf16')
    assert gen.gru_tiling() == 1
    other = HipEngine(P.pr, w, n_streams=8192, mfcc_precision='f32', gru_precision='bf16', ring_precision='bf16')
    dev = torch.device('cuda', 0)
    side = torch.cuda.Stream(device=dev)
    opcm = torch.from_numpy(synth.batch_pcm(1, 1)[0, 0]).to(dev)[None, :].expand(8192, 1024).contiguous()
    oout = torch.zeros(8192, device=dev)
    stop = threading.Event()
    launched = [0]

    def background():
        while not stop.is_set():
            for _ in range(20):
                other.update_device(opcm.data_ptr(), 1024, oout.data_ptr(), side.cuda_stream)
            launched[0] += 20
            side.synchronize()

    th = threading.Thread(target=background)
    th.start()
    try:
        bad_out, bad_win = _position_soak(gen, 120)
    finally:
        stop.set()
        th.join()
    side.synchronize()
    same_other = bool((oout == oout[0]).all().item())
    gen.close(); other.close()
    assert launched[0] >= 20 and same_other
    assert (bad_out, bad_win) == (0, 0), 'general float32 frames disagree between positions: %d outputs, %d windows' % (bad_out, bad_win)


@pytest.mark.parametrize('dup', ['push', 'keep'])
def test_colliding_mel_grid_either_duplicate_rule(dup):
    """40 filters over the 257 bins of n_fft = 512: the mel grid starts 0, 0, 1, 2 ... -- the one setting family where the two
    possible behaviours of sonopy 0.1.2's correct_grid differ.  `mel_filterbank(..., duplicates=)` builds either table; the
    kernels serve both, offline and streaming, against the oracle switched the same way (sonopy_restated.DUPLICATES)."""
    import warnings
    from mycroft_precise_amd import vectorization as V
    from mycroft_precise_amd._lib import HipEngine
    from oracle import sonopy_restated as sr
    kw = dict(n_filt=40)
    hpr = P.pr.copy()
    hpr.__dict__.update(kw)
    opr = ol.Params(**kw)
    w = synth.make_weights(seed=11)
    n, n_up = 19, 40
    pcm = _stream_batch(['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet'], n_up)
    prev_o, prev_p = sr.DUPLICATES, V.sonopy_duplicates
    try:
        sr.DUPLICATES = dup
        V.sonopy_duplicates = dup
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', V.UnverifiedFilterbank)
            bank = V.mel_filterbank(16000, 40, 257)
            assert np.array_equal(bank, sr.filterbanks(16000, 40, 257))
            eng = HipEngine(hpr, w, n_streams=n)                 # (the module default reaches pe_create through HipEngine)
            explicit = HipEngine(hpr, w, n_streams=n, mel_filters=V.mel_filterbank(16000, 40, 257, duplicates=dup))
        audio = synth.stream_pcm(2, 20000).astype(np.float64) / 32768.0
        want = sr.mfcc_spec(audio, 16000, (opr.window_samples, opr.hop_samples), 512, 40, 13)
        assert np.abs(eng.vectorize_raw(audio) - want).max() <= 1e-9
        ref = ol.BatchedOracle(w, n, opr)
        for u in range(n_up):
            got = eng.update(pcm[u])
            assert np.array_equal(got, explicit.update(pcm[u]))
            assert np.abs(got.astype(np.float64) - ref.update_raw(pcm[u])).max() <= GUARD_RAW, u
        assert np.abs(eng.get_vectors().astype(np.float64) - ref.mfccs).max() <= TOL_FEAT32
        eng.close(); explicit.close()
    finally:
        sr.DUPLICATES, V.sonopy_duplicates = prev_o, prev_p


def test_random_chunk_sizes_streams_and_call_shapes(stock_weights):
    """Seeded sweep over what the frame tasks' addressing depends on: chunk length (even: dword sample pairs through
    bounds-checked buffer loads; odd: the sample-by-sample path; shorter than a frame; longer than a window, which
    cannot be fused), batch size (ragged tiles, waves with idle lanes) and updates per call (frames that straddle
    chunk rows of different updates).  Every raw probability and the final features against the oracle."""
    from mycroft_precise_amd.network_runner import BatchedListener
    rng = np.random.default_rng(20260924)
    kinds_all = ['tone_noise', 'tone_noise', 'quiet', 'square', 'zeros']
    for case in range(14):
        chunk = int(rng.choice([rng.integers(60, 500), rng.integers(500, 1100), rng.integers(1100, 4000)]))
        if case % 3 == 0:
            chunk &= ~1                                            # make sure the aligned fast path gets its share
        n = int(rng.integers(1, 70))
        depth = int(rng.choice([1, 1, 2, 5]))
        n_up = 12 * depth
        kinds = [kinds_all[int(k)] for k in rng.integers(0, len(kinds_all), n)]
        pcm = _stream_batch(kinds, n_up, chunk)
        hip = BatchedListener(stock_weights, n)
        ref = ol.BatchedOracle(stock_weights, n)
        if depth > 1:
            hip.engine.reserve_updates(depth, chunk)
        for u in range(0, n_up, depth):
            want = np.stack([ref.update_raw(pcm[u + i]) for i in range(depth)])
            got = hip.engine.update_many(pcm[u:u + depth]) if depth > 1 else hip.update_raw(pcm[u])[None]
            assert np.abs(got.astype(np.float64) - want).max() <= GUARD_RAW, (case, chunk, n, depth, u)
        assert np.abs(hip.engine.get_vectors().astype(np.float64) - ref.mfccs).max() <= TOL_FEAT32, (case, chunk, n, depth)
        hip.engine.close()


def test_waves_owning_more_than_64_streams():
    """mfcc_frame_tasks keeps the counters of 64 streams per wave in registers and refills them batch by batch when a
    wave owns more: with one resident frame workgroup per compute unit (PE_FRAME_WG_PER_CU=1: 1024 waves) that starts
    above 65536 streams.  tests/many_streams_check.py: oracle on seeded streams, identical input => identical output
    in either batch of a wave, fused == two launches, at 66000 streams."""
    env = dict(os.environ, PE_FRAME_WG_PER_CU='1', PYTHONPATH=REPO + os.pathsep + os.environ.get('PYTHONPATH', ''))
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tests', 'many_streams_check.py'), '66000'], env=env, cwd=REPO,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.startswith('ok'), (out.stdout[-500:], out.stderr[-2000:])


def test_host_fed_pipeline_equals_synchronous_updates(stock_weights):
    """pe_update_async / pe_wait (the host-fed path: pinned staging ring, next chunk crossing PCIe under the running update):
    bit-identical to pe_update whatever the mix of pinned (zero-copy) and pageable buffers, with more updates enqueued
    than the ring holds, with synchronous entry points in between (they wait for the updates in flight), after a clear."""
    from mycroft_precise_amd.network_runner import BatchedListener
    n, n_up = 70, 30
    kinds = ['tone_noise'] * (n - 3) + ['zeros', 'square', 'quiet']
    pcm = _stream_batch(kinds, n_up)
    ref = BatchedListener(stock_weights, n)
    want = np.stack([ref.update_raw(pcm[u]) for u in range(n_up)])
    hip = BatchedListener(stock_weights, n)
    bufs = [hip.chunk_buffer(1024) for _ in range(4)]
    pinned_out = hip.engine.host_array((n_up, n), np.float32)
    outs = []
    for u in range(n_up):
        if u % 3 == 0:                                    # pageable in, pageable out
            outs.append(hip.update_raw_async(pcm[u].copy()))
        else:                                             # pinned in (rotating), pinned out
            b = bufs[u % 4]
            b[:] = pcm[u]
            outs.append(hip.update_raw_async(b, pinned_out[u]))
        if u == 11:                                       # a synchronous entry point in the middle: sees updates 0..11
            feats = hip.engine.get_vectors()
            r2 = BatchedListener(stock_weights, n)
            for v in range(12):
                r2.update_raw(pcm[v])
            assert np.array_equal(feats, r2.engine.get_vectors())
    hip.wait()
    assert np.array_equal(np.stack(outs), want)
    # mixed with the synchronous call and a clear
    hip.clear()
    a = hip.update_raw_async(pcm[0])
    b = hip.update_raw(pcm[1])                             # waits for the one in flight first
    assert np.array_equal(a, want[0]) and np.array_equal(b, want[1])
    with pytest.raises(ValueError):
        hip.engine.update_async(pcm[0], np.empty(n + 1, np.float32))
    with pytest.raises(EOFError):
        hip.update_raw_async(np.empty((n, 0), '<i2'))


# ---- BASELINE configs[0]: one stream through the engine executable (plumbing) ------------------------
def test_precise_engine_subprocess_protocol(model_file, stock_weights):
    """PreciseEngine + `python -m mycroft_precise_amd.scripts.engine`: raw int16 on stdin, one ASCII
    float per chunk on stdout (runner.py:54-67 <-> engine.py:53-63)."""
    from mycroft_precise_amd.runner import PreciseEngine
    g = golden('listener_chunk2048.npz')
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get('PYTHONPATH', ''))
    eng = PreciseEngine([sys.executable, '-m', 'mycroft_precise_amd.scripts.engine'], model_file, 2048)
    eng.proc = subprocess.Popen(eng.exe_args, stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, cwd=REPO)
    try:
        data = g['pcm'][0].tobytes()
        got = [eng.get_prediction(data[off:off + 2048]) for off in range(0, 2048 * 12, 2048)]
    finally:
        eng.stop()
    assert _decode_flips(got, g['decoded'][0][:12]) == DECODE_FLIPS['engine']


def test_reference_module_name_runs_the_engine(model_file, stock_weights):
    """`python -m precise.scripts.engine model 2048` with compat/ on the path -- the literal command the reference's
    PreciseEngine spawns (runner.py:48-52): under -m the alias file IS __main__, so it has to start the engine itself
    (it used to exit 0 without a line, and a PreciseEngine saw immediate EOF)."""
    from mycroft_precise_amd.runner import PreciseEngine
    g = golden('listener_chunk2048.npz')
    env = dict(os.environ, PYTHONPATH=os.path.join(REPO, 'compat') + os.pathsep + REPO + os.pathsep + os.environ.get('PYTHONPATH', ''))
    eng = PreciseEngine([sys.executable, '-m', 'precise.scripts.engine'], model_file, 2048)
    eng.proc = subprocess.Popen(eng.exe_args, stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, cwd=REPO)
    try:
        data = g['pcm'][0].tobytes()
        got = [eng.get_prediction(data[off:off + 2048]) for off in range(0, 2048 * 6, 2048)]
    finally:
        eng.stop()
    assert _decode_flips(got, g['decoded'][0][:6]) == DECODE_FLIPS['engine']
