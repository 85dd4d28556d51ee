"""
Audio -> prediction: drop-in for ``precise.network_runner``
(/root/reference/precise/network_runner.py:31-153) with the arithmetic on an MI355X.

Same surface as the reference -- ``Runner`` (plug-in ABC), ``Listener(model_name, chunk_size=-1,
runner_cls=None)`` with ``update`` / ``update_vectors`` / ``clear`` / ``find_runner`` and the
``pr`` / ``mfccs`` / ``window_audio`` / ``runner`` / ``threshold_decoder`` attributes -- plus
``BatchedListener``, the same thing for B lock-step streams (the natural unit on a GPU).

Where the work happens:
  * leftover-PCM bookkeeping, framing, MFCC and the [T x F] feature window: ``pe_update*``
    (csrc/mfcc_wave_device.h, mfcc_device.h; any other ListenerParams: mfcc_general_device.h) -- the state lives in
    HBM, not in numpy arrays;
  * the network: ``HipRunner`` -> ``pe_predict`` / fused into ``pe_update`` (csrc/gru_device.h, gru_cw_device.h, ...);
  * ``ThresholdDecoder.decode``: host, one float64 per prediction, as in the reference.
"""
from abc import ABCMeta, abstractmethod
from os.path import splitext

import numpy as np

from ._lib import HipEngine
from .model import load_weights
from .params import inject_params, pr, Vectorizer
from .threshold_decoder import ThresholdDecoder
from .util import pcm16_from
from .vectorization import add_deltas, vectorize_raw


class Runner(metaclass=ABCMeta):
    """Executes a trained model on vectorized audio (network_runner.py:31-42)."""

    @abstractmethod
    def predict(self, inputs: np.ndarray) -> np.ndarray:
        """[N, T, F] -> [N, 1]"""

    @abstractmethod
    def run(self, inp: np.ndarray) -> float:
        """[T, F] -> scalar"""


def _require_streamable(params):
    """The streaming kernels carry MFCC rows (<= 16 coefficients per frame) of the sonopy (``mfccs``) or the
    legacy speechpy (``speechpy_mfccs``) front end; mel rows (n_filt = 20 wide, Vectorizer.mels) exist in the
    offline form only -- the reference's own Listener cannot stream them either (network_runner.py:104,144)."""
    if params.vectorizer not in (Vectorizer.mfccs, Vectorizer.speechpy_mfccs):
        raise NotImplementedError('Vectorizer.%s cannot be streamed on the device: Vectorizer.mfccs and '
                                  'Vectorizer.speechpy_mfccs have streaming kernels (mels: vectorization.vectorize '
                                  '/ vectorize_raw)' % {Vectorizer.mels: 'mels'}.get(params.vectorizer, str(params.vectorizer)))


def _engine_params(use_delta=None):
    _require_streamable(pr)
    snap = pr.copy()
    if use_delta is not None:
        snap.__dict__['use_delta'] = use_delta
    return snap


class HipRunner(Runner):
    """The network on the GPU.  ``runner_cls(model_name)`` signature as the reference's runners
    (network_runner.py:47,79); ``weights`` may be handed over directly instead of a file."""

    def __init__(self, model_name: str = None, weights: dict = None, n_streams: int = 1, device: int = 0,
                 mfcc_precision: str = 'f64'):
        if weights is None:
            weights = load_weights(model_name)
        self.model_name = model_name
        self.weights = weights
        self.engine = HipEngine(_engine_params(), weights, n_streams=n_streams, device=device,
                                mfcc_precision=mfcc_precision)

    def predict(self, inputs: np.ndarray) -> np.ndarray:
        return self.engine.predict(inputs)

    def run(self, inp: np.ndarray) -> float:
        return self.predict(np.asarray(inp)[np.newaxis])[0][0]

    def evaluate(self, audio: np.ndarray, chunk_size: int = 4096) -> np.ndarray:
        """The reference's offline batch evaluation (precise/scripts/simulate.py:92-104) in one device
        call: MFCC of the whole recording, one prediction every ``chunk_size // hop_samples`` frames
        over the sliding ``n_features``-frame window.  -> raw outputs [N, 1]."""
        hops = int(chunk_size) // pr.hop_samples
        if hops < 1:
            raise ValueError('chunk_size must be at least hop_samples (%d)' % pr.hop_samples)
        return self.engine.evaluate(audio, hops)


# The reference's two runner names (network_runner.py:45-95): `from precise.network_runner import Listener, KerasRunner`
# (scripts/train_incremental.py:45) and `TensorFlowRunner` resolve to the one implementation here, whatever the file format.
KerasRunner = TensorFlowRunner = HipRunner


def _placeholder_weights(n_in):
    return {'gru': [(np.zeros((n_in, 3), np.float32), np.zeros((1, 3), np.float32), np.zeros(3, np.float32))],
            'dense_kernel': np.zeros((1, 1), np.float32), 'dense_bias': np.zeros(1, np.float32)}


class Listener:
    """Preprocesses one audio stream into MFCC vectors and executes the network
    (network_runner.py:98-153)."""

    def __init__(self, model_name: str, chunk_size: int = -1, runner_cls: type = None):
        self.pr = inject_params(model_name)
        self.chunk_size = chunk_size
        runner_cls = runner_cls or self.find_runner(model_name)
        self.threshold_decoder = ThresholdDecoder(self.pr.threshold_config, pr.threshold_center)
        self.window_audio = np.array([])
        self._mfccs = None
        self._engine = None
        self._front = None          # MFCC-only engine behind a foreign runner, created on first need
        self._float_mode = False
        self.runner = runner_cls(model_name)
        self.clear()

    @property
    def runner(self):
        return self._runner

    @runner.setter
    def runner(self, runner):
        """The reference lets callers swap the runner of a live Listener (scripts/train_incremental.py:87-88):
        predictions must come from the new runner from the next update on, the stream state stays."""
        self._runner = runner
        fused = isinstance(runner, HipRunner) and runner.engine.n_streams == 1
        if fused:
            engine = runner.engine
        else:       # a foreign Runner plugged into the reference's seam: the GPU still does the MFCC
            if self._front is None:
                self._front = HipEngine(_engine_params(use_delta=False), _placeholder_weights(self.pr.n_mfcc))
            engine = self._front
        old = self._engine
        self._fused, self._engine = fused, engine
        if old is not None and old is not engine and not getattr(self, '_float_mode', False):
            # carry the stream over: the feature window as it stands, then the leftover samples
            engine.set_vectors(old.get_vectors())
            if len(self.window_audio):
                left = np.rint(np.asarray(self.window_audio, dtype=np.float64) * 32768.0).astype('<i2')
                engine.update_vectors(left.reshape(1, -1), want_features=False)

    @staticmethod
    def find_runner(model_name: str):
        runners = {'.npz': HipRunner, '.net': HipRunner, '.pb': HipRunner}
        ext = splitext(model_name)[-1]
        if ext not in runners:
            raise ValueError('File extension of ' + model_name + ' must be: ' + str(list(runners)))
        return runners[ext]

    def clear(self):
        self.window_audio = np.array([])
        self._engine.clear()
        self._mfccs = np.zeros((self.pr.n_features, self.pr.n_mfcc))
        self._float_mode = False

    @property
    def mfccs(self) -> np.ndarray:
        """The [n_features, n_mfcc] feature window (float64 view of the device's float32 ring)."""
        if self._mfccs is None:
            self._mfccs = self._engine.get_vectors()[0].astype(np.float64)
        return self._mfccs

    @mfccs.setter
    def mfccs(self, value):
        """``listener.mfccs = window`` works on the reference's plain attribute (network_runner.py:104) and leaves
        ``window_audio`` alone: here the window is installed on the device too (``pe_set_vectors``) and the leftover
        samples are handed back to it, exactly as when the runner is replaced after construction."""
        window = np.asarray(value, dtype=np.float64).reshape(self.pr.n_features, self.pr.n_mfcc)
        self._mfccs = window.copy()
        if self._float_mode:
            return                                     # float samples: the window lives on the host (see _update_vectors_float)
        self._engine.set_vectors(window[np.newaxis].astype(np.float32))
        if len(self.window_audio):
            left = np.rint(np.asarray(self.window_audio, dtype=np.float64) * 32768.0).astype('<i2')
            self._engine.update_vectors(left.reshape(1, -1), want_features=False)

    def _read(self, stream):
        """What the reference appends to ``window_audio`` (network_runner.py:126-137) as a pair
        (int16 PCM or None, float samples): bytes and file-like streams are little-endian int16 PCM
        (``buffer_to_audio``); an ndarray IS the samples, whatever its dtype -- it is PCM for the device path
        only when every sample is exactly k/32768."""
        if isinstance(stream, np.ndarray):
            audio = stream.reshape(-1)
            scaled = audio.astype(np.float64) * 32768.0
            if scaled.size == 0 or (np.all(np.rint(scaled) == scaled) and scaled.min() >= -32768 and scaled.max() <= 32767):
                return scaled.astype('<i2'), audio
            return None, audio
        chunk = stream if isinstance(stream, (bytes, bytearray)) else stream.read(self.chunk_size)
        if len(chunk) == 0:
            raise EOFError
        pcm = pcm16_from(chunk)
        return pcm, pcm.astype(np.float32) / np.float32(32768.0)

    def _frames_in(self, n: int) -> int:
        """Frames the vectorizer returns for ``n`` buffered samples (speechpy drops the last full window)."""
        if n < self.pr.window_samples:
            return 0
        full = 1 + (n - self.pr.window_samples) // self.pr.hop_samples
        return full - 1 if self.pr.vectorizer == Vectorizer.speechpy_mfccs else full

    def _track_leftover(self, audio: np.ndarray):
        # host mirror of the reference's ``window_audio`` attribute (sample bookkeeping only)
        self.window_audio = np.concatenate((self.window_audio, audio))
        self.window_audio = self.window_audio[self._frames_in(len(self.window_audio)) * self.pr.hop_samples:]

    def _update_vectors_float(self, audio: np.ndarray) -> np.ndarray:
        """Arbitrary float samples (scaled, resampled or mixed audio, e.g. ``load_audio`` output, which is
        k/32767): the reference's own bookkeeping (network_runner.py:137-144) on the host, float64 like its
        ``window_audio``; the MFCC of the buffered samples is the device's stateless kernel (``vectorize_raw``).
        Once a stream has taken such samples its leftover is no longer int16, so it stays on this path until
        ``clear()``."""
        if not self._float_mode:
            self._mfccs = self.mfccs                   # pull the device's feature window over once
            self._float_mode = True
        self.window_audio = np.concatenate((self.window_audio, audio))
        if len(self.window_audio) >= self.pr.window_samples:
            new = vectorize_raw(self.window_audio)
            self.window_audio = self.window_audio[len(new) * self.pr.hop_samples:]
            if len(new) > len(self._mfccs):
                new = new[-len(self._mfccs):]
            self._mfccs = np.concatenate((self._mfccs[len(new):], new))
        return self._mfccs

    def update_vectors(self, stream) -> np.ndarray:
        pcm, audio = self._read(stream)
        if self._float_mode or pcm is None:
            return self._update_vectors_float(audio)
        if pcm.size == 0:                              # an empty ndarray: nothing to append (no EOFError, :126-127)
            return self.mfccs
        self._track_leftover(audio)
        self._mfccs = self._engine.update_vectors(pcm.reshape(1, -1))[0].astype(np.float64)
        return self._mfccs

    def _update_raw32(self, stream):
        """The network output as the runner hands it over: a numpy float32 scalar on the HIP path, exactly
        what ``TensorFlowRunner.run`` returns in the reference (network_runner.py:73-74)."""
        pcm, audio = self._read(stream)
        if self._fused and not self._float_mode and pcm is not None and pcm.size:
            self._track_leftover(audio)
            self._mfccs = None                     # fetched from the device on demand
            return self._engine.update(pcm.reshape(1, -1))[0]
        mfccs = self.update_vectors(audio if pcm is None or self._float_mode or not pcm.size else pcm.astype(np.float32) / np.float32(32768.0))
        if self.pr.use_delta:
            mfccs = add_deltas(mfccs)
        return self.runner.run(mfccs)

    def update_raw(self, stream) -> float:
        """``update`` without the ThresholdDecoder: the raw network output."""
        return float(self._update_raw32(stream))

    def update(self, stream) -> float:
        # the decoder sees the runner's own scalar type: with a float32 network output the reference's
        # ``1 / x - 1`` (functions.py:99-101) is float32 arithmetic, and the table bin follows from that
        return float(self.threshold_decoder.decode(self._update_raw32(stream)))


class BatchedListener:
    """
    ``Listener`` for ``n_streams`` independent streams on one device.  ``update(chunks)`` takes one equal-sized chunk for
    EVERY stream ([n_streams, chunk_samples] int16, or a list of bytes objects) and returns one prediction per stream;
    ``update(chunks, streams=ids)`` takes chunks for the named streams only ([len(ids), chunk_samples]) and leaves every other
    stream exactly as it is -- each stream then advances at its own pace, as one reference ``Listener`` per client does
    (network_runner.py:125-146), and the call costs what its active streams cost.  Stream state never leaves HBM.
    """

    def __init__(self, model, n_streams: int, device: int = 0, mfcc_precision: str = 'f64', params=None,
                 gru_precision: str = 'f32', ring_precision: str = 'f32'):
        if isinstance(model, str):
            self.pr = inject_params(model).copy()
            weights = load_weights(model)
        else:
            self.pr = (params or pr).copy()
            weights = model
        self.n_streams = int(n_streams)
        self.weights = weights
        _require_streamable(self.pr)
        self.engine = HipEngine(self.pr, weights, n_streams=self.n_streams, device=device,
                                mfcc_precision=mfcc_precision, gru_precision=gru_precision, ring_precision=ring_precision)
        self.threshold_decoder = ThresholdDecoder(self.pr.threshold_config, self.pr.threshold_center)
        self.engine.set_decoder(self.threshold_decoder)
        self._trigger = False

    def set_trigger(self, chunk_size: int = 2048, sensitivity: float = 0.5, trigger_level: int = 3):
        """One TriggerDetector (runner/precise_runner/runner.py:115-142) per stream, on the device;
        ``chunk_size`` in bytes as in the reference."""
        self.engine.set_trigger(chunk_size, sensitivity, trigger_level)
        self._trigger = True

    def _pcm(self, chunks, n_rows=None) -> np.ndarray:
        n_rows = self.n_streams if n_rows is None else n_rows
        if isinstance(chunks, np.ndarray):
            pcm = chunks
        else:
            rows = [pcm16_from(c) for c in chunks]
            if len({r.size for r in rows}) > 1:
                raise ValueError('all streams of one call must supply equal-sized chunks (one call per chunk length)')
            pcm = np.stack(rows) if rows else np.empty((0, 0), dtype='<i2')
        if pcm.ndim != 2 or pcm.shape[0] != n_rows:
            raise ValueError('expected [%d, chunk_samples] int16, got %r' % (n_rows, pcm.shape))
        if pcm.shape[1] == 0 and n_rows > 0:
            raise EOFError
        return pcm

    def _ids(self, streams) -> np.ndarray:
        ids = np.asarray(streams, dtype=np.int64).reshape(-1)
        if ids.size and (ids.min() < 0 or ids.max() >= self.n_streams):
            raise ValueError('stream ids must be in 0..%d' % (self.n_streams - 1))
        if np.unique(ids).size != ids.size:
            raise ValueError('a stream may be named once per call')
        return ids.astype(np.int32)

    def clear(self, mask=None):
        self.engine.clear(mask)

    def update_vectors(self, chunks) -> np.ndarray:
        """-> [n_streams, n_features, n_mfcc] float32"""
        return self.engine.update_vectors(self._pcm(chunks))

    def update_raw(self, chunks, streams=None) -> np.ndarray:
        """-> raw network outputs float32 [n_streams]; with ``streams=ids``: [len(ids)], in the order of ``ids``"""
        if streams is None:
            return self.engine.update(self._pcm(chunks))
        ids = self._ids(streams)
        if ids.size == 0:
            return np.empty(0, dtype=np.float32)
        return self.engine.update_subset(ids, self._pcm(chunks, ids.size))

    def update(self, chunks, streams=None) -> np.ndarray:
        """-> decoded confidences float64 [n_streams] (ThresholdDecoder.decode on the device); with ``streams=ids`` the
        confidences of those streams only (decoded on the host, one value each, as the reference's Listener.update does)"""
        if streams is None:
            return self.engine.decode(self.update_raw(chunks))
        raw = self.update_raw(chunks, streams)
        return np.array([self.threshold_decoder.decode(r) for r in raw], dtype=np.float64)

    # -- host-fed pipeline: what a server that receives audio over the network does (scripts/engine.py:60-63 per stream) --
    def chunk_buffer(self, chunk_samples: int = 1024) -> np.ndarray:
        """A pinned [n_streams, chunk_samples] int16 array to receive audio into: ``update_raw_async`` reads it in place
        (no staging copy), which is what reaches PCIe line rate.  Keep a few and rotate: a buffer is read until ``wait()``
        or until three more updates have been enqueued."""
        return self.engine.host_array((self.n_streams, int(chunk_samples)), '<i2')

    def update_raw_async(self, chunks, out: np.ndarray = None) -> np.ndarray:
        """Enqueue one update; returns the float32 [n_streams] array its raw outputs land in after ``wait()``.  Up to three
        updates are in flight: chunk u + 1 crosses PCIe while update u runs.  Same bits as ``update_raw``."""
        return self.engine.update_async(self._pcm(chunks), out)

    def wait(self):
        self.engine.wait()

    def update_detect(self, chunks):
        """-> (confidences float64 [n_streams], activations bool [n_streams]); needs set_trigger()."""
        if not self._trigger:
            raise RuntimeError('call set_trigger() first')
        return self.engine.decode(self.update_raw(chunks), want_fired=True)
