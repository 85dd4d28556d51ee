#!/usr/bin/env python3
"""
Generate tests/golden/*.npz by running THE REFERENCE'S OWN, UNMODIFIED code
(/root/reference/precise/{network_runner,vectorization,params,util,threshold_decoder}.py)
in the build container.  TEST INFRASTRUCTURE ONLY -- runs here, never on the GPU box
(/root/reference does not exist there); the committed .npz files are what travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

The reference's two third-party arithmetic dependencies are absent, so they are supplied
through the reference's own seams:
  * ``sonopy``  -> sys.modules['sonopy'] = oracle.sonopy_restated   (vectorization.py:24)
  * the network -> ``Listener(..., runner_cls=<numpy Keras-GRU restatement>)``
                                                                  (network_runner.py:101,106)
Everything else that executes -- Listener framing/leftover/ring logic, buffer_to_audio,
vectorize/add_deltas, ListenerParams, ThresholdDecoder -- is reference code.
"""
import os
import sys

sys.dont_write_bytecode = True          # never write __pycache__ into /root/reference
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference')
sys.path.insert(0, '/root/reference/runner')

import warnings
import numpy as np

from oracle import sonopy_restated, speechpy_restated, keras_gru
from mycroft_precise_amd import synth

# Third-party arithmetic: the REAL packages win whenever they are importable (the day a maintainer runs this
# with sonopy 0.1.2 / speechpy-fast 2.4 / Keras 2.2.4 installed, the fixtures upgrade from "restated" to
# "pinned" without a code change); otherwise the restatements are plugged into the reference's own seams.
# Which one was used is written into every .npz as ``provenance``.
PROVENANCE = {}


def _third_party(name, restated):
    try:
        mod = __import__(name)
        PROVENANCE[name] = 'real %s %s' % (name, getattr(mod, '__version__', '?'))
        return mod
    except ImportError:
        sys.modules[name] = restated
        PROVENANCE[name] = 'restated (oracle/%s.py), real package not importable' % restated.__name__.split('.')[-1]
        return restated


_third_party('sonopy', sonopy_restated)
_third_party('speechpy', speechpy_restated)
try:
    import keras                                            # noqa: F401
    PROVENANCE['keras'] = 'real keras %s: gen_keras_runner() writes keras_runner.npz from the reference\'s own KerasRunner' % keras.__version__
    HAVE_KERAS = True
except ImportError:
    HAVE_KERAS = False
    PROVENANCE['keras'] = 'restated (oracle/keras_gru.py via the runner_cls seam), real Keras/TensorFlow not importable'
PROVENANCE['glue'] = 'reference code, unmodified: /root/reference/precise/{network_runner,vectorization,params,util,threshold_decoder,functions}.py, runner/precise_runner/runner.py'
PROVENANCE['numpy'] = np.__version__


def provenance():
    return np.array(['%s: %s' % kv for kv in sorted(PROVENANCE.items())])


def save(name, **arrays):
    np.savez_compressed(os.path.join(OUT, name), provenance=provenance(), **arrays)


warnings.simplefilter('ignore', DeprecationWarning)        # np.fromstring in util.py:37

from precise.network_runner import Listener                # noqa: E402  (reference code)
from precise.threshold_decoder import ThresholdDecoder     # noqa: E402
from precise.vectorization import vectorize, vectorize_raw, add_deltas   # noqa: E402
from precise.util import buffer_to_audio                   # noqa: E402
from precise.params import pr                              # noqa: E402
from precise_runner.runner import TriggerDetector, ReadWriteStream   # noqa: E402  (reference code)

OUT = os.path.join(REPO, 'tests', 'golden')


class RecordingRunner(keras_gru.NumpyRunner):
    last_raw = None

    def run(self, inp):
        raw = super().run(inp)
        type(self).last_raw = raw
        return raw


def run_listener(pcm: np.ndarray, chunk_bytes: int, weights):
    """Feed ``pcm`` (int16 1-D) through the reference Listener in chunk_bytes pieces.
    Returns raw outputs, decoded outputs, feature ring after every update, leftover length."""
    cls = type('R', (RecordingRunner,), {'weights': weights})
    listener = Listener('synthetic-model-not-on-disk', chunk_bytes, runner_cls=cls)
    data = pcm.tobytes()
    raws, decs, rings, left = [], [], [], []
    for off in range(0, len(data) - chunk_bytes + 1, chunk_bytes):
        dec = listener.update(data[off:off + chunk_bytes])
        raws.append(cls.last_raw)
        decs.append(dec)
        rings.append(listener.mfccs.copy())
        left.append(len(listener.window_audio))
    return (np.array(raws, dtype=np.float32), np.array(decs, dtype=np.float64),
            np.array(rings, dtype=np.float64), np.array(left, dtype=np.int64))


def gen_mels():
    """Vectorizer.mels through the reference's own dispatch (vectorization.py:31-35,46-50,62-84): the
    process-global ``pr`` is switched to the mels vectorizer, so feature rows are n_filt wide."""
    from precise.params import Vectorizer
    import precise.params as rp
    saved = rp.pr.vectorizer
    try:
        rp.pr.__dict__['vectorizer'] = Vectorizer.mels
        assert rp.pr.feature_size == rp.pr.n_filt
        vz = {}
        for name, n in (('short', 5000), ('long', 40000), ('one_window', 1600)):
            a = synth.stream_pcm(23, n, 'tone_noise').astype(np.float32) / np.float32(32768.0)
            vz['audio_' + name] = a
            vz['vec_' + name] = vectorize(a)
            vz['raw_' + name] = vectorize_raw(a)
        z = np.zeros(4000, dtype=np.float32)                       # eps clip in every filter
        vz['audio_zeros'], vz['raw_zeros'] = z, vectorize_raw(z)
        save('vectorize_mels.npz', **vz)
    finally:
        rp.pr.__dict__['vectorizer'] = saved


def gen_speechpy(weights):
    """The legacy vectorizer (vectorization.py:40-42) through the reference's own dispatch and its own Listener:
    ``pr.vectorizer = speechpy_mfccs`` is what ``inject_params`` sets for every .params file without a
    ``vectorizer`` key (params.py:147,155)."""
    from precise.params import Vectorizer
    import precise.params as rp
    saved = rp.pr.vectorizer
    try:
        rp.pr.__dict__['vectorizer'] = Vectorizer.speechpy_mfccs
        vz = {}
        for name, n in (('short', 5000), ('exact', 24000), ('long', 40000), ('one_window', 1600), ('window_plus_hop', 2400)):
            a = synth.stream_pcm(31, n, 'tone_noise').astype(np.float32) / np.float32(32768.0)
            vz['audio_' + name] = a
            vz['vec_' + name] = vectorize(a)
            vz['raw_' + name] = vectorize_raw(a)
        z = np.zeros(4000, dtype=np.float32)                       # zero_handling in every filter and the energy
        vz['audio_zeros'], vz['raw_zeros'] = z, vectorize_raw(z)
        cases = [('tone_noise', 2), ('zeros', 0), ('square', 12), ('quiet', 1)]
        n_updates = 40
        pcm_all, raw_all, dec_all, ring_last, left_all, ring_u7 = [], [], [], [], [], []
        for kind, s_ in cases:
            pcm = synth.stream_pcm(s_, n_updates * 1024, kind)
            raws, decs, rings, left = run_listener(pcm, 2048, weights)
            pcm_all.append(pcm); raw_all.append(raws); dec_all.append(decs)
            ring_last.append(rings[-1]); ring_u7.append(rings[7]); left_all.append(left)
        pcm = synth.stream_pcm(7, 48000, 'tone_noise')
        for cb in (1000, 3200, 6400, 96000):
            raws, decs, rings, left = run_listener(pcm, cb, weights)
            vz['odd_raw_%d' % cb], vz['odd_ring_last_%d' % cb], vz['odd_leftover_%d' % cb] = raws, rings[-1], left
        save('speechpy.npz', kinds=np.array([k for k, _ in cases]), pcm=np.array(pcm_all), raw=np.array(raw_all),
             decoded=np.array(dec_all), ring_last=np.array(ring_last), ring_u7=np.array(ring_u7),
             leftover=np.array(left_all), odd_pcm=pcm, **vz)
    finally:
        rp.pr.__dict__['vectorizer'] = saved


def gen_float_audio(weights):
    """Listener.update(ndarray) with arbitrary float samples (network_runner.py:126-127): load_audio-style
    k/32767 values (util.py:45-65) and a mixed / rescaled signal, chunked as scripts/train_incremental.py does."""
    cls = type('R', (RecordingRunner,), {'weights': weights})
    out = {}
    pcm = synth.stream_pcm(17, 30 * 1024, 'tone_noise')
    for name, audio in (('div32767', pcm.astype(np.float32) / np.float32(32767.0)),
                        ('mixed64', 0.37 * pcm.astype(np.float64) / 32768.0 + 0.01 * np.sin(np.arange(pcm.size) * 0.05))):
        listener = Listener('synthetic-model-not-on-disk', 2048, runner_cls=cls)
        raws, decs = [], []
        for off in range(0, len(audio) - 1023, 1024):
            decs.append(listener.update(audio[off:off + 1024]))
            raws.append(cls.last_raw)
        out['audio_' + name] = audio
        out['raw_' + name] = np.array(raws, dtype=np.float32)
        out['decoded_' + name] = np.array(decs, dtype=np.float64)
        out['ring_last_' + name] = listener.mfccs.copy()
        out['leftover_' + name] = len(listener.window_audio)
    # a stream that starts as PCM bytes and continues with float samples
    listener = Listener('synthetic-model-not-on-disk', 2048, runner_cls=cls)
    audio = pcm.astype(np.float32) / np.float32(32767.0)
    raws = []
    for u in range(30):
        if u < 12:
            listener.update(pcm[u * 1024:(u + 1) * 1024].tobytes())
        else:
            listener.update(audio[u * 1024:(u + 1) * 1024])
        raws.append(cls.last_raw)
    out['pcm'] = pcm
    out['raw_bytes_then_float'] = np.array(raws, dtype=np.float32)
    save('listener_float_audio.npz', **out)


def gen_keras_runner(weights):
    """Only where the real Keras is importable (never in the offline build container): the reference's own model
    builder and KerasRunner (/root/reference/precise/model.py:57-91, network_runner.py:77-95) on the synthetic weights
    -- the fixture that pins oracle/keras_gru.py (K1-K8) to real Keras bits.  tests/test_oracle.py and the GPU suite
    pick the file up when it exists."""
    import tempfile
    from precise.model import create_model, ModelParams          # reference code
    from precise.network_runner import KerasRunner                # reference code
    k, rk, b = weights['gru'][0]
    model = create_model(None, ModelParams(skip_acc=True))
    model.set_weights([k, rk, b, weights['dense_kernel'], weights['dense_bias']])
    rng = np.random.default_rng(11)
    x = rng.normal(0.0, 3.0, (64, 29, k.shape[0])).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'synthetic.net')
        model.save(path)
        y = KerasRunner(path).predict(x)
    save('keras_runner.npz', inputs=x, outputs=np.asarray(y, np.float32))


def main():
    os.makedirs(OUT, exist_ok=True)
    if '--mels-only' in sys.argv:
        gen_mels()
        return
    gen_mels()
    gen_speechpy(synth.make_weights())
    gen_float_audio(synth.make_weights())
    weights = synth.make_weights()

    # --- params.py derived sizes -------------------------------------------------------
    save('params_default.npz',
             window_samples=pr.window_samples, hop_samples=pr.hop_samples,
             buffer_samples=pr.buffer_samples, n_features=pr.n_features,
             max_samples=pr.max_samples, feature_size=pr.feature_size,
             n_fft=pr.n_fft, n_filt=pr.n_filt, n_mfcc=pr.n_mfcc)

    # --- util.buffer_to_audio ----------------------------------------------------------
    edge = np.array([0, 1, -1, 32767, -32768, 12345, -12345, 256, -256], dtype='<i2')
    save('buffer_to_audio.npz', pcm=edge, audio=buffer_to_audio(edge.tobytes()))

    # --- Listener.update streaming: 1024-sample chunks (runner.py:48 default) ------------
    cases = [('tone_noise', 0), ('tone_noise', 5), ('tone_noise', 96), ('zeros', 0),
             ('square', 12), ('square', 3), ('quiet', 1)]
    n_updates = 40
    pcm_all, raw_all, dec_all, ring_last, left_all, ring_u7 = [], [], [], [], [], []
    for kind, s in cases:
        pcm = synth.stream_pcm(s, n_updates * 1024, kind)
        raws, decs, rings, left = run_listener(pcm, 2048, weights)
        pcm_all.append(pcm); raw_all.append(raws); dec_all.append(decs)
        ring_last.append(rings[-1]); ring_u7.append(rings[7]); left_all.append(left)
    save('listener_chunk2048.npz',
                        kinds=np.array([k for k, _ in cases]), streams=np.array([s for _, s in cases]),
                        pcm=np.array(pcm_all), raw=np.array(raw_all), decoded=np.array(dec_all),
                        ring_last=np.array(ring_last), ring_u7=np.array(ring_u7),
                        leftover=np.array(left_all))

    # --- other chunk sizes (bytes): 1000, 3200, 6400 and one whole-buffer update ---------
    odd = {}
    pcm = synth.stream_pcm(7, 48000, 'tone_noise')
    for cb in (1000, 3200, 6400, 20000):
        raws, decs, rings, left = run_listener(pcm, cb, weights)
        odd['raw_%d' % cb] = raws
        odd['decoded_%d' % cb] = decs
        odd['ring_last_%d' % cb] = rings[-1]
        odd['leftover_%d' % cb] = left
    # 96000 bytes in ONE update: > n_features new frames at once (network_runner.py:142-143)
    raws, decs, rings, left = run_listener(pcm, 96000, weights)
    odd['raw_96000'], odd['decoded_96000'] = raws, decs
    odd['ring_last_96000'], odd['leftover_96000'] = rings[-1], left
    save('listener_oddchunks.npz', pcm=pcm, **odd)

    # --- vectorize / vectorize_raw / add_deltas (vectorization.py:46-84) ------------------
    vz = {}
    for name, n in (('short', 5000), ('exact', 24000), ('long', 40000), ('one_window', 1600)):
        a = synth.stream_pcm(11, n, 'tone_noise').astype(np.float32) / np.float32(32768.0)
        vz['audio_' + name] = a
        vz['vec_' + name] = vectorize(a)
    a = synth.stream_pcm(11, 8000, 'tone_noise').astype(np.float32) / np.float32(32768.0)
    vz['raw_feats_8000'] = vectorize_raw(a)
    vz['deltas_8000'] = add_deltas(vz['raw_feats_8000'])
    save('vectorize.npz', **vz)

    # --- ThresholdDecoder (threshold_decoder.py:38-70) -----------------------------------
    td = {}
    grid = np.concatenate([[0.0, 1.0], np.linspace(1e-6, 1 - 1e-6, 197),
                           1 / (1 + np.exp(-np.linspace(-30, 30, 121)))])
    thr = np.linspace(0.01, 0.99, 50)
    for name, cfg, center in (('default', ((6, 4),), 0.2), ('two', ((-3.0, 2.0), (4.0, 1.5)), 0.5),
                              ('narrow', ((0.0, 0.05),), 0.3)):
        d = ThresholdDecoder(cfg, center)
        td['cfg_' + name] = np.array(cfg, dtype=np.float64)
        td['center_' + name] = center
        td['decode_' + name] = np.array([d.decode(float(v)) for v in grid])
        # what Listener.update hands the decoder is the runner's numpy float32 scalar (network_runner.py:73-74,
        # 152-153): `1 / x - 1` (functions.py:99-101) is then float32 arithmetic
        td['decode32_' + name] = np.array([d.decode(v) for v in grid.astype(np.float32)], dtype=np.float64)
        td['encode_' + name] = np.array([d.encode(float(v)) for v in thr])
        td['cd_len_' + name] = len(d.cd)
        td['min_out_' + name] = d.min_out
        td['out_range_' + name] = d.out_range
    save('threshold_decoder.npz', grid=grid, thr=thr, **td)

    # --- precise_runner TriggerDetector / ReadWriteStream (runner/precise_runner/runner.py:76-142) ----
    rng = np.random.default_rng(7)
    tr = {}
    for name, chunk_size, sens, level in (('default', 2048, 0.5, 3), ('small', 1024, 0.8, 1),
                                          ('big', 8192, 0.2, 5), ('lvl0', 4096, 0.5, 0)):
        # bursts of high probability separated by lulls, plus noise
        probs = np.clip(rng.random(400) * 0.6 + (np.sin(np.arange(400) / 7.0) > 0.3) * 0.5, 0, 1)
        det = TriggerDetector(chunk_size, sens, level)
        fired, act = [], []
        for p_ in probs:
            fired.append(det.update(float(p_)))
            act.append(det.activation)
        tr['probs_' + name] = probs
        tr['cfg_' + name] = np.array([chunk_size, sens, level], dtype=np.float64)
        tr['fired_' + name] = np.array(fired, dtype=np.bool_)
        tr['activation_' + name] = np.array(act, dtype=np.int64)
    # every read carries a timeout: the reference's read() blocks forever when a chop leaves
    # fewer bytes than requested
    s = ReadWriteStream(b'0123456789abcde', chop_samples=10)      # 15 % 10 = 5 bytes survive the chop
    s.write(b'FGHIJKLM')                                           # 23 % 10 = 3 survive
    tr['rws_chop_read'] = np.frombuffer(s.read(2, timeout=0.2), dtype=np.uint8)
    tr['rws_chop_left'] = np.frombuffer(s.read(1, timeout=0.2), dtype=np.uint8)
    tr['rws_chop_short'] = np.frombuffer(s.read(5, timeout=0.05), dtype=np.uint8)   # times out -> b''
    s = ReadWriteStream(b'0123456789abcdef', chop_samples=8)      # len % chop == 0: nothing is dropped
    tr['rws_nochop_read'] = np.frombuffer(s.read(3, timeout=0.2), dtype=np.uint8)
    tr['rws_nochop_len'] = len(s)
    save('precise_runner.npz', **tr)

    if HAVE_KERAS:
        gen_keras_runner(weights)

    # --- weights used by every fixture -----------------------------------------------------
    k, rk, b = weights['gru'][0]
    save('weights_stock_seed42.npz', kernel=k, recurrent_kernel=rk,
                        bias=b, dense_kernel=weights['dense_kernel'], dense_bias=weights['dense_bias'])
    print('golden fixtures written to', OUT)
    for f in sorted(os.listdir(OUT)):
        print('  %-32s %8d bytes' % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == '__main__':
    main()
