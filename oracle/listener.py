"""
CPU restatement of the reference's streaming glue.  TEST INFRASTRUCTURE ONLY.

Follows, function by function:
  * ``buffer_to_audio``            /root/reference/precise/util.py:35-37
  * ``ListenerParams`` derived sizes /root/reference/precise/params.py:73-109, defaults :140-144
  * ``vectorize_raw`` / ``vectorize`` / ``add_deltas``
                                   /root/reference/precise/vectorization.py:46-84
  * ``Listener.__init__/clear/update_vectors/update``
                                   /root/reference/precise/network_runner.py:98-153
  * ``ThresholdDecoder``           /root/reference/precise/threshold_decoder.py:38-70
    with ``sigmoid/asigmoid/pdf``  /root/reference/precise/functions.py:94-108

Pinned by ``tests/golden/*.npz`` (outputs of the reference's own unmodified classes, see
``oracle/gen_golden.py``); the MFCC / GRU arithmetic plugged in underneath is the
restated third-party code of ``sonopy_restated.py`` / ``keras_gru.py`` (parity unpinned).
"""
from math import exp, log, sqrt, pi, floor

import numpy as np

from . import sonopy_restated as sonopy
from . import speechpy_restated as speechpy
from . import keras_gru


class Params:
    """params.py:28-118 (only what the hot path reads)."""

    def __init__(self, buffer_t=1.5, window_t=0.1, hop_t=0.05, sample_rate=16000, sample_depth=2,
                 n_fft=512, n_filt=20, n_mfcc=13, use_delta=False, vectorizer=2,
                 threshold_config=((6, 4),), threshold_center=0.2):
        self.buffer_t, self.window_t, self.hop_t = buffer_t, window_t, hop_t
        self.sample_rate, self.sample_depth = sample_rate, sample_depth
        self.n_fft, self.n_filt, self.n_mfcc = n_fft, n_filt, n_mfcc
        self.use_delta, self.vectorizer = use_delta, vectorizer
        self.threshold_config, self.threshold_center = threshold_config, threshold_center

    @property
    def window_samples(self):                     # params.py:84-87
        return int(self.sample_rate * self.window_t + 0.5)

    @property
    def hop_samples(self):                        # params.py:89-92
        return int(self.sample_rate * self.hop_t + 0.5)

    @property
    def buffer_samples(self):                     # params.py:73-77
        samples = int(self.sample_rate * self.buffer_t + 0.5)
        return self.hop_samples * (samples // self.hop_samples)

    @property
    def n_features(self):                         # params.py:79-82
        return 1 + int(floor((self.buffer_samples - self.window_samples) / self.hop_samples))

    @property
    def max_samples(self):                        # params.py:94-97
        return int(self.buffer_t * self.sample_rate)

    @property
    def feature_size(self):                       # params.py:99-109
        n = self.n_filt if self.vectorizer == 1 else self.n_mfcc
        return 2 * n if self.use_delta else n


def buffer_to_audio(buffer: bytes) -> np.ndarray:
    """util.py:35-37: little-endian int16 -> float32 / 32768.0"""
    return np.frombuffer(buffer, dtype='<i2').astype(np.float32, order='C') / np.float32(32768.0)


def vectorize_raw(audio, pr: Params):
    """vectorization.py:46-50 with the ``Vectorizer.mfccs`` entry (:36-39), or -- ``pr.vectorizer == 3`` -- the
    legacy ``Vectorizer.speechpy_mfccs`` entry (:40-42)."""
    if len(audio) == 0:
        raise ValueError('Cannot vectorize empty audio!')
    if pr.vectorizer == 3:
        return speechpy.feature.mfcc(np.asarray(audio), pr.sample_rate, pr.window_t, pr.hop_t, pr.n_mfcc, pr.n_filt, pr.n_fft)
    return sonopy.mfcc_spec(audio, pr.sample_rate, (pr.window_samples, pr.hop_samples),
                            num_filt=pr.n_filt, fft_size=pr.n_fft, num_coeffs=pr.n_mfcc)


def add_deltas(features):
    """vectorization.py:53-59"""
    deltas = np.zeros_like(features)
    deltas[1:] = features[1:] - features[:-1]
    return np.concatenate([features, deltas], -1)


def vectorize(audio, pr: Params):
    """vectorization.py:62-84: last max_samples -> frames -> left zero pad / tail crop."""
    if len(audio) > pr.max_samples:
        audio = audio[-pr.max_samples:]
    feats = vectorize_raw(audio, pr)
    if len(feats) < pr.n_features:
        feats = np.concatenate([np.zeros((pr.n_features - len(feats), feats.shape[1])), feats])
    if len(feats) > pr.n_features:
        feats = feats[-pr.n_features:]
    return feats


class ThresholdDecoder:
    """threshold_decoder.py:38-70"""

    def __init__(self, mu_stds, center=0.5, resolution=200, min_z=-4, max_z=4):
        self.min_out = int(min(mu + min_z * std for mu, std in mu_stds))
        self.max_out = int(max(mu + max_z * std for mu, std in mu_stds))
        self.out_range = self.max_out - self.min_out
        pts = np.linspace(self.min_out, self.max_out, resolution * self.out_range)
        pd = np.zeros_like(pts)
        for mu, std in mu_stds:
            if std != 0:
                pd = pd + (1.0 / (std * sqrt(2 * pi))) * np.exp(-(pts - mu) ** 2 / (2 * std ** 2))
        self.cd = np.cumsum(pd / (resolution * len(mu_stds)))
        self.center = center

    def decode(self, raw_output: float) -> float:
        if raw_output == 1.0 or raw_output == 0.0:
            return raw_output
        if self.out_range == 0:
            cp = int(raw_output > self.min_out)
        else:
            ratio = (-log(1 / raw_output - 1) - self.min_out) / self.out_range
            ratio = min(max(ratio, 0.0), 1.0)
            cp = self.cd[int(ratio * (len(self.cd) - 1) + 0.5)]
        if cp < self.center:
            return 0.5 * cp / self.center
        return 0.5 + 0.5 * (cp - self.center) / (1 - self.center)

    def encode(self, threshold: float) -> float:
        threshold = 0.5 * threshold / self.center
        if threshold < 0.5:
            cp = threshold * self.center * 2
        else:
            cp = (threshold - 0.5) * 2 * (1 - self.center) + self.center
        ratio = np.searchsorted(self.cd, cp) / len(self.cd)
        return 1 / (1 + exp(-(self.min_out + self.out_range * ratio)))


class OracleListener:
    """Single-stream restatement of network_runner.py:98-153 (one stream, any chunk size)."""

    def __init__(self, weights, pr: Params = None, chunk_size: int = -1):
        self.pr = pr or Params()
        self.weights = weights
        self.chunk_size = chunk_size
        self.threshold_decoder = ThresholdDecoder(self.pr.threshold_config, self.pr.threshold_center)
        self.clear()

    def clear(self):                              # :121-123
        self.window_audio = np.array([])
        self.mfccs = np.zeros((self.pr.n_features, self.pr.n_mfcc))

    def update_vectors(self, stream):             # :125-146
        if isinstance(stream, np.ndarray):
            buffer_audio = stream
        else:
            chunk = stream if isinstance(stream, (bytes, bytearray)) else stream.read(self.chunk_size)
            if len(chunk) == 0:
                raise EOFError
            buffer_audio = buffer_to_audio(chunk)
        self.window_audio = np.concatenate((self.window_audio, buffer_audio))
        if len(self.window_audio) >= self.pr.window_samples:
            new = vectorize_raw(self.window_audio, self.pr)
            self.window_audio = self.window_audio[len(new) * self.pr.hop_samples:]
            if len(new) > len(self.mfccs):
                new = new[-len(self.mfccs):]
            self.mfccs = np.concatenate((self.mfccs[len(new):], new))
        return self.mfccs

    def update_raw32(self, stream):
        """update() up to, not including, the ThresholdDecoder: the runner's numpy float32 scalar (:152)."""
        mfccs = self.update_vectors(stream)
        if self.pr.use_delta:
            mfccs = add_deltas(mfccs)
        return keras_gru.predict(mfccs[np.newaxis], self.weights)[0][0]

    def update_raw(self, stream) -> float:
        return float(self.update_raw32(stream))

    def update(self, stream) -> float:            # :148-153: the decoder is handed the float32 scalar
        return self.threshold_decoder.decode(self.update_raw32(stream))


class BatchedOracle:
    """
    The same arithmetic vectorised over B lock-step streams (all streams receive equal-sized
    chunks and were cleared together, so their leftover lengths agree).  This is the
    ``cpu_baseline`` ("port") of bench.py and the fast checker for the -m gpu parity tests.
    State per stream exactly as the reference: leftover audio (float64) + [T, F] feature ring.
    """

    def __init__(self, weights, n_streams: int, pr: Params = None):
        self.pr = pr or Params()
        self.weights = weights
        self.n = n_streams
        self.clear()

    def clear(self):
        self.window_audio = np.zeros((self.n, 0))
        self.mfccs = np.zeros((self.n, self.pr.n_features, self.pr.n_mfcc))

    def update_vectors(self, pcm: np.ndarray) -> np.ndarray:
        """pcm: int16 [B, chunk_samples] -> features [B, T, F] float64"""
        pr = self.pr
        audio = pcm.astype(np.float32) / np.float32(32768.0)              # util.py:35-37
        self.window_audio = np.concatenate((self.window_audio, audio.astype(np.float64)), axis=1)
        length = self.window_audio.shape[1]
        if length >= pr.window_samples:
            starts = sonopy.frame_starts(length, pr.window_samples, pr.hop_samples)
            if pr.vectorizer == 3:
                starts = starts[:speechpy.n_frames(length, pr.window_samples, pr.hop_samples)]     # S1: one fewer
            idx = starts[:, None] + np.arange(pr.n_fft)[None, :]            # Q2 / S2 crop
            front = speechpy.mfcc_from_frames if pr.vectorizer == 3 else sonopy.mfcc_from_frames
            new = front(self.window_audio[:, idx], pr.sample_rate, pr.n_fft, pr.n_filt, pr.n_mfcc)   # [B, n, F]
            n_new = new.shape[1]
            self.window_audio = self.window_audio[:, n_new * pr.hop_samples:]
            if n_new > pr.n_features:
                new = new[:, -pr.n_features:]
            self.mfccs = np.concatenate((self.mfccs[:, new.shape[1]:], new), axis=1)
        return self.mfccs

    def update_raw(self, pcm: np.ndarray) -> np.ndarray:
        """-> raw network outputs [B] float32"""
        feats = self.update_vectors(pcm)
        if self.pr.use_delta:
            feats = np.stack([add_deltas(f) for f in feats])
        return keras_gru.predict(feats, self.weights)[:, 0]
