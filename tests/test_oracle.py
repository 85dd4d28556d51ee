"""
CPU: the oracle (restatement) against the golden fixtures produced by the reference's own
unmodified code (oracle/gen_golden.py).  This is what pins the oracle's glue.
"""
import warnings

import numpy as np
import pytest

from conftest import golden
from oracle import listener as ol
from oracle import sonopy_restated as so
from mycroft_precise_amd import synth


def test_params_derived_sizes():
    g = golden('params_default.npz')
    pr = ol.Params()
    for k in ('window_samples', 'hop_samples', 'buffer_samples', 'n_features', 'max_samples',
              'feature_size', 'n_fft', 'n_filt', 'n_mfcc'):
        assert getattr(pr, k) == int(g[k]), k
    assert (pr.window_samples, pr.hop_samples, pr.n_features) == (1600, 800, 29)


def test_buffer_to_audio_bit_exact():
    g = golden('buffer_to_audio.npz')
    out = ol.buffer_to_audio(g['pcm'].tobytes())
    assert out.dtype == np.float32
    assert np.array_equal(out, g['audio'])


def test_weights_fixture_matches_generator(stock_weights):
    w = synth.make_weights()
    for a, b in zip(w['gru'][0], stock_weights['gru'][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(w['dense_kernel'], stock_weights['dense_kernel'])


def test_listener_stream_chunk2048(stock_weights):
    g = golden('listener_chunk2048.npz')
    for i in range(len(g['streams'])):
        lis = ol.OracleListener(stock_weights)
        data = g['pcm'][i].tobytes()
        raws, decs = [], []
        for u, off in enumerate(range(0, len(data), 2048)):
            raw = lis.update_raw32(data[off:off + 2048])
            raws.append(raw)
            decs.append(lis.threshold_decoder.decode(raw))
            assert len(lis.window_audio) == g['leftover'][i][u]
            if u == 7:
                assert np.array_equal(lis.mfccs, g['ring_u7'][i])
        assert np.array_equal(lis.mfccs, g['ring_last'][i])
        assert np.array_equal(np.array(raws, dtype=np.float32), g['raw'][i])
        assert np.array_equal(np.array(decs), g['decoded'][i])


@pytest.mark.parametrize('cb', [1000, 3200, 6400, 20000, 96000])
def test_listener_odd_chunk_sizes(stock_weights, cb):
    g = golden('listener_oddchunks.npz')
    data = g['pcm'].tobytes()
    lis = ol.OracleListener(stock_weights)
    raws, decs, left = [], [], []
    for off in range(0, len(data) - cb + 1, cb):
        raws.append(lis.update_raw32(data[off:off + cb]))
        decs.append(lis.threshold_decoder.decode(raws[-1]))
        left.append(len(lis.window_audio))
    assert np.array_equal(np.array(raws, dtype=np.float32), g['raw_%d' % cb])
    assert np.array_equal(np.array(decs), g['decoded_%d' % cb])
    assert np.array_equal(lis.mfccs, g['ring_last_%d' % cb])
    assert np.array_equal(np.array(left), g['leftover_%d' % cb])


def test_empty_chunk_is_eof(stock_weights):
    with pytest.raises(EOFError):
        ol.OracleListener(stock_weights).update(b'')


def test_vectorize_pad_crop_and_deltas():
    g = golden('vectorize.npz')
    pr = ol.Params()
    for name in ('short', 'exact', 'long', 'one_window'):
        v = ol.vectorize(g['audio_' + name], pr)
        assert v.shape == (29, 13)
        assert np.array_equal(v, g['vec_' + name]), name
    assert np.array_equal(ol.add_deltas(g['raw_feats_8000']), g['deltas_8000'])
    with pytest.raises(ValueError):
        ol.vectorize_raw(np.array([]), pr)


def test_mels_vectorizer_matches_reference_dispatch():
    """Vectorizer.mels (vectorization.py:32-35): the fixture was produced by the reference's own
    vectorize / vectorize_raw with pr.vectorizer = mels; rows are n_filt wide, no DCT."""
    from oracle import sonopy_restated as so
    g = golden('vectorize_mels.npz')
    pr = ol.Params()
    for name in ('short', 'long', 'one_window', 'zeros'):
        raw = so.mel_spec(g['audio_' + name], pr.sample_rate, (pr.window_samples, pr.hop_samples),
                          num_filt=pr.n_filt, fft_size=pr.n_fft)
        assert raw.shape[1] == pr.n_filt
        assert np.array_equal(raw, g['raw_' + name]), name
    assert np.all(g['raw_zeros'] == np.log(np.finfo(float).eps))


@pytest.mark.parametrize('name', ['default', 'two', 'narrow'])
def test_threshold_decoder(name):
    g = golden('threshold_decoder.npz')
    d = ol.ThresholdDecoder([tuple(r) for r in g['cfg_' + name]], float(g['center_' + name]))
    assert len(d.cd) == int(g['cd_len_' + name])
    assert d.min_out == int(g['min_out_' + name]) and d.out_range == int(g['out_range_' + name])
    dec = np.array([d.decode(float(v)) for v in g['grid']])
    assert np.array_equal(dec, g['decode_' + name])
    # the scalar type the runners return (numpy float32): `1 / x - 1` is then float32 arithmetic
    dec32 = np.array([d.decode(v) for v in g['grid'].astype(np.float32)])
    assert np.array_equal(dec32, g['decode32_' + name])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        enc = np.array([d.encode(float(v)) for v in g['thr']])
    assert np.array_equal(enc, g['encode_' + name], equal_nan=True)


def test_batched_oracle_equals_single_stream(stock_weights):
    """BatchedOracle (cpu_baseline / GPU checker) == the pinned single-stream restatement."""
    n_up, streams = 36, [0, 5, 96]
    pcm = np.stack([synth.stream_pcm(s, n_up * 1024).reshape(n_up, 1024) for s in streams], axis=1)
    bo = ol.BatchedOracle(stock_weights, len(streams))
    singles = [ol.OracleListener(stock_weights) for _ in streams]
    for u in range(n_up):
        raw_b = bo.update_raw(pcm[u])
        for j, lis in enumerate(singles):
            raw_s = lis.update_raw(pcm[u, j].tobytes())
            assert abs(raw_s - raw_b[j]) < 2e-6
            assert np.allclose(lis.mfccs, bo.mfccs[j], rtol=0, atol=1e-10)


def test_filterbank_shape_and_support():
    fb = so.filterbanks(16000, 20, 257)
    assert fb.shape == (20, 257)
    assert int((fb != 0).sum()) == 455
    assert fb.min() >= 0.0 and fb.max() == 1.0
    # every bin feeds at most two filters (triangles only overlap with their neighbours)
    assert int((fb != 0).sum(0).max()) <= 2


def test_frame_crop_quirk_q2():
    """Only the first n_fft samples of each 1600-sample window reach the FFT."""
    rng = np.random.default_rng(0)
    a = rng.normal(size=4000)
    b = a.copy()
    b[512:800] = 7.0          # inside window 0, beyond the crop, before window 1
    b[800 + 512:1600] = -3.0
    assert np.array_equal(so.mfcc_spec(a, 16000, (1600, 800)), so.mfcc_spec(b, 16000, (1600, 800)))


# ---- legacy speechpy vectorizer (vectorization.py:40-42; params.py:147,155) ---------------------------------------
def test_speechpy_vectorize_matches_reference_dispatch():
    g = golden('speechpy.npz')
    pr = ol.Params(vectorizer=3)
    for name in ('short', 'exact', 'long', 'one_window', 'window_plus_hop'):
        assert np.array_equal(ol.vectorize_raw(g['audio_' + name], pr), g['raw_' + name]), name
        assert np.array_equal(ol.vectorize(g['audio_' + name], pr), g['vec_' + name]), name
    assert g['raw_one_window'].shape == (0, 13)           # S1: exactly one window yields no frame
    assert g['raw_window_plus_hop'].shape == (1, 13)
    assert np.array_equal(ol.vectorize_raw(g['audio_zeros'], pr), g['raw_zeros'])


def test_speechpy_listener_streams(stock_weights):
    g = golden('speechpy.npz')
    pr = ol.Params(vectorizer=3)
    for i in range(len(g['kinds'])):
        lis = ol.OracleListener(stock_weights, pr)
        data = g['pcm'][i].tobytes()
        raws, decs = [], []
        for u, off in enumerate(range(0, len(data), 2048)):
            raws.append(lis.update_raw32(data[off:off + 2048]))
            decs.append(lis.threshold_decoder.decode(raws[-1]))
            assert len(lis.window_audio) == g['leftover'][i][u]
            if u == 7:
                assert np.array_equal(lis.mfccs, g['ring_u7'][i])
        assert np.array_equal(lis.mfccs, g['ring_last'][i])
        assert np.array_equal(np.array(raws, dtype=np.float32), g['raw'][i])
        assert np.array_equal(np.array(decs), g['decoded'][i])
    data = g['odd_pcm'].tobytes()
    for cb in (1000, 3200, 6400, 96000):
        lis = ol.OracleListener(stock_weights, pr)
        raws, left = [], []
        for off in range(0, len(data) - cb + 1, cb):
            raws.append(lis.update_raw32(data[off:off + cb]))
            left.append(len(lis.window_audio))
        assert np.array_equal(np.array(raws, dtype=np.float32), g['odd_raw_%d' % cb])
        assert np.array_equal(lis.mfccs, g['odd_ring_last_%d' % cb])
        assert np.array_equal(np.array(left), g['odd_leftover_%d' % cb])


def test_speechpy_batched_oracle_equals_single_stream(stock_weights):
    pr = ol.Params(vectorizer=3)
    n_up = 36
    pcm = synth.batch_pcm(3, n_up)
    bo = ol.BatchedOracle(stock_weights, 3, pr)
    singles = [ol.OracleListener(stock_weights, pr) for _ in range(3)]
    for u in range(n_up):
        raw_b = bo.update_raw(pcm[u])
        for j, lis in enumerate(singles):
            assert abs(lis.update_raw(pcm[u, j].tobytes()) - raw_b[j]) < 2e-6
            assert np.allclose(lis.mfccs, bo.mfccs[j], rtol=0, atol=1e-10)


# ---- arbitrary float samples into Listener.update (network_runner.py:126-127) --------------------------------------
def test_listener_float_ndarray_audio(stock_weights):
    g = golden('listener_float_audio.npz')
    for name in ('div32767', 'mixed64'):
        audio = g['audio_' + name]
        lis = ol.OracleListener(stock_weights)
        raws = [lis.update_raw32(audio[off:off + 1024]) for off in range(0, len(audio) - 1023, 1024)]
        assert np.array_equal(np.array(raws, dtype=np.float32), g['raw_' + name])
        assert np.array_equal(lis.mfccs, g['ring_last_' + name])
        assert len(lis.window_audio) == int(g['leftover_' + name])


def test_golden_fixtures_carry_their_provenance(capsys):
    """Every fixture says which third-party arithmetic produced it (real package or restatement)."""
    import os
    from conftest import GOLDEN
    for f in sorted(os.listdir(GOLDEN)):
        g = golden(f)
        prov = [str(x) for x in g['provenance']]
        assert any(p.startswith('sonopy:') for p in prov) and any(p.startswith('keras:') for p in prov), f
        assert any(p.startswith('glue: reference code, unmodified') for p in prov), f
        print(f, '|', '; '.join(p for p in prov if not p.startswith('glue')))


def test_keras_restatement_against_a_real_keras_fixture_when_present():
    """tests/golden/keras_runner.npz exists only where oracle/gen_golden.py ran with the real Keras importable (it is
    the reference's own KerasRunner on the synthetic weights).  When it is there, the restated GRU must match it."""
    import os
    from conftest import REPO
    GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')
    path = os.path.join(GOLDEN_DIR, 'keras_runner.npz')
    if not os.path.isfile(path):
        pytest.skip('no real-Keras fixture on this machine (Keras / TensorFlow are not installable offline)')
    from oracle import keras_gru
    g, w = np.load(path), np.load(os.path.join(GOLDEN_DIR, 'weights_stock_seed42.npz'))
    weights = {'gru': [(w['kernel'], w['recurrent_kernel'], w['bias'])], 'dense_kernel': w['dense_kernel'], 'dense_bias': w['dense_bias']}
    assert np.abs(keras_gru.predict(g['inputs'], weights) - g['outputs']).max() <= 1e-6


def test_three_bf16_pieces_carry_a_float32_product(stock_weights):
    """The arithmetic claim behind pe_set_gru_tiling 2 (csrc/gru_x3_device.h), restated in numpy: a float32 value is the exact
    sum of three bf16 pieces (round to nearest even of the running remainder), and the six piece products
    w_hi v_hi + w_hi v_mid + w_mid v_hi + w_mid v_mid + w_hi v_lo + w_lo v_hi stand for the float32 product to within one
    float32 rounding -- so a GRU window evaluated that way sits as close to a float64 evaluation as the float32 oracle does."""
    from oracle import keras_gru

    def to_bf16(v):                      # round to nearest even, as v_cvt_pk_bf16_f32 does
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)).view(np.float32)

    def split3(v):
        r, out = v.astype(np.float32), []
        for _ in range(3):
            p = to_bf16(r)
            out.append(p)
            r = (r - p).astype(np.float32)          # exact
        return out

    rng = np.random.default_rng(7)
    v = np.concatenate([rng.normal(0, 3, 5000), rng.normal(0, 1e-3, 5000), rng.uniform(-40, 40, 5000)]).astype(np.float32)
    hi, mid, lo = split3(v)
    assert np.array_equal((hi.astype(np.float64) + mid + lo).astype(np.float32), v)                   # the split is exact
    w = rng.normal(0, 0.5, v.shape).astype(np.float32)
    wp, vp = split3(w), [hi, mid, lo]
    six = sum(wp[i].astype(np.float64) * vp[j] for i, j in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)))
    exact = w.astype(np.float64) * v
    assert np.all(np.abs(six - exact) <= 2.0 ** -22 * np.abs(exact) + 1e-300)                          # dropped terms: < 2^-22 |w v|

    # a whole window: stock network, 29 x 13 features of realistic size
    g = stock_weights['gru'][0]
    W, U, b = g
    H = U.shape[0]
    x = (rng.standard_normal((512, 29, 13)) * 4).astype(np.float32)
    x[:, :, 0] = rng.uniform(-25, 10, (512, 29))
    terms = ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0))

    def mm(a, B):
        ap, Bp = split3(a), split3(B)
        acc = np.zeros((a.shape[0], B.shape[1]), np.float32)
        for i, j in terms:
            acc = (acc + ap[i].astype(np.float64) @ Bp[j].astype(np.float64)).astype(np.float32)
        return acc

    h = np.zeros((512, H), np.float32)
    for t in range(29):
        ax = mm(x[:, t], W) + b
        ah = mm(h, U[:, :2 * H])
        z = keras_gru.hard_sigmoid((ax[:, :H] + ah[:, :H]).astype(np.float32))
        r = keras_gru.hard_sigmoid((ax[:, H:2 * H] + ah[:, H:]).astype(np.float32))
        hh = (ax[:, 2 * H:] + mm((r * h).astype(np.float32), U[:, 2 * H:])).astype(np.float32)
        h = (z * h + (np.float32(1) - z) * hh).astype(np.float32)
    logit = h @ stock_weights['dense_kernel'] + stock_weights['dense_bias']
    p_x3 = (1.0 / (1.0 + np.exp(-logit.astype(np.float64))))[:, 0]
    p32 = keras_gru.predict(x, stock_weights)[:, 0].astype(np.float64)
    p64 = keras_gru.predict(x, stock_weights, dtype=np.float64)[:, 0]
    d_x3, d_32 = np.abs(p_x3 - p64).max(), np.abs(p32 - p64).max()
    assert d_x3 <= 2 * d_32 + 1e-6, (d_x3, d_32)
    assert np.abs(p_x3 - p32).max() <= 2e-5                                                             # the float32 guard of the GPU tests


def test_torch_batched_restatement_equals_the_numpy_oracle():
    """oracle/torch_batched.py (bench.py's second CPU baseline, BASELINE.md B2 "numpy / torch-CPU") against the numpy oracle it
    restates: features to float64 round-off, probabilities to float32 summation order -- stock and wide networks."""
    torch = pytest.importorskip('torch')
    from oracle.torch_batched import TorchBatchedOracle
    from mycroft_precise_amd import synth
    for units in ((20,), (48, 40)):
        w = synth.make_weights(units=units, seed=3)
        B, n_up = 24, 40
        pcm = synth.batch_pcm(B, n_up)
        a, b = ol.BatchedOracle(w, B), TorchBatchedOracle(w, B)
        for u in range(n_up):
            pa, pb = a.update_raw(pcm[u]), b.update_raw(pcm[u])
            assert np.abs(pa.astype(np.float64) - pb).max() <= 1e-6, (units, u)
        assert np.abs(a.mfccs - b.mfccs.numpy()).max() <= 1e-9
