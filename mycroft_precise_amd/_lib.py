"""
ctypes binding of libprecise_engine.so (C ABI: include/precise_engine.h).

There is NO CPU fallback: if the HIP library is missing or a call fails this module raises.
"""
import ctypes as C
import importlib.util
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# PE_LIB: load another build of the library (an A/B against an older tree) instead of the in-tree one
LIB_PATH = os.environ.get('PE_LIB') or os.path.join(HERE, 'libprecise_engine.so')

PE_OK, PE_ERR_INVALID, PE_ERR_HIP, PE_ERR_UNSUPPORTED, PE_ERR_NOMEM, PE_ERR_EOF = range(6)
ABI_VERSION = 7


class PeParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'sample_rate', 'window_samples', 'hop_samples', 'n_fft', 'n_filt', 'n_mfcc', 'n_features',
        'use_delta', 'mfcc_precision', 'gru_precision', 'vectorizer', 'ring_precision')]


class PeGruLayer(C.Structure):
    _fields_ = [('n_in', C.c_int32), ('units', C.c_int32),
                ('kernel', C.POINTER(C.c_float)), ('recurrent_kernel', C.POINTER(C.c_float)),
                ('bias', C.POINTER(C.c_float))]


class PeWeights(C.Structure):
    _fields_ = [('n_layers', C.c_int32), ('layers', C.POINTER(PeGruLayer)),
                ('dense_kernel', C.POINTER(C.c_float)), ('dense_bias', C.c_float)]


class PeInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'n_streams', 'n_features', 'n_mfcc', 'units', 'n_layers', 'ring_slots', 'carry_capacity',
        'mfcc_precision', 'gru_precision')] + [('device_bytes', C.c_int64)]


EXPORTS = {
    # name: (restype, argtypes)
    'pe_abi_version': (C.c_int, []),
    'pe_last_global_error': (C.c_char_p, []),
    'pe_create': (C.c_int, [C.POINTER(PeParams), C.POINTER(C.c_double), C.POINTER(PeWeights), C.c_int32,
                            C.c_int32, C.POINTER(C.c_void_p)]),
    'pe_destroy': (C.c_int, [C.c_void_p]),
    'pe_last_error': (C.c_char_p, [C.c_void_p]),
    'pe_clear': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pe_update': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'pe_update_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'pe_update_device_keep': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'pe_update_subset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    'pe_update_subset_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'pe_set_renumber_at': (C.c_int, [C.c_void_p, C.c_uint32]),
    'pe_host_alloc': (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    'pe_host_free': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pe_update_async': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'pe_wait': (C.c_int, [C.c_void_p]),
    'pe_reserve_updates': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    'pe_update_many': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'pe_update_many_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'pe_update_vectors': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'pe_update_vectors_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'pe_get_vectors': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pe_set_vectors': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pe_run_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'pe_predict': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'pe_predict_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'pe_vectorize_raw': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                   C.POINTER(C.c_int64)]),
    'pe_vectorize_mels': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                   C.POINTER(C.c_int64)]),
    'pe_evaluate': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                              C.POINTER(C.c_int64)]),
    'pe_set_decoder': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double]),
    'pe_set_trigger': (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32]),
    'pe_decode_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pe_decode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pe_get_info': (C.c_int, [C.c_void_p, C.POINTER(PeInfo)]),
    'pe_get_stream_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pe_set_fused': (C.c_int, [C.c_void_p, C.c_int32]),
    'pe_set_gru_waves': (C.c_int, [C.c_void_p, C.c_int32]),
    'pe_set_gru_tiling': (C.c_int, [C.c_void_p, C.c_int32]),
    'pe_get_gru_tiling': (C.c_int, [C.c_void_p]),
    'pe_set_input_projection': (C.c_int, [C.c_void_p, C.c_int32]),
    'pe_set_timing': (C.c_int, [C.c_void_p, C.c_int32]),
    'pe_get_last_timing': (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
}

_lib = None


class HipLibraryMissing(ImportError):
    pass


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process do not share devices pointers or streams, so when torch is installed
    its copy is loaded first and libprecise_engine.so binds to it -- whatever the import order."""
    if os.environ.get('PRECISE_AMD_HIP_RUNTIME', '') == 'system':
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load (once) and return the bound library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            'libprecise_engine.so is not built (expected at %s). Build it with '
            '`python -m mycroft_precise_amd._build`; there is no CPU fallback.' % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)       # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pe_abi_version() != ABI_VERSION:
        raise HipLibraryMissing('libprecise_engine.so ABI %d != expected %d; rebuild' %
                                (lib.pe_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('precise_engine error %d: %s' % (code, msg))
        self.code = code


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class HipEngine:
    """
    One C-ABI engine: the streaming state of ``n_streams`` audio streams plus one network on one
    MI355X.  Thin, allocation-free wrapper; the reference-shaped classes live in
    ``network_runner.py``.
    """

    def __init__(self, params, weights, n_streams=1, device=0, mfcc_precision='f64', mel_filters=None,
                 gru_precision='f32', ring_precision='f32'):
        from .vectorization import mel_filterbank, speechpy_filterbank
        from .params import Vectorizer
        self._lib = load()
        self._h = C.c_void_p()
        self._async_keep = []
        self._views = 0                # host_array() buffers still referenced by numpy arrays (their memory dies with the engine)
        self._close_pending = False
        self.n_streams = int(n_streams)
        self.n_features = int(params.n_features)
        self.n_mfcc = int(params.n_mfcc)
        self.n_filt = int(params.n_filt)
        self.feature_size = int(params.n_mfcc) * (2 if params.use_delta else 1)
        prec = {'f64': 0, 'f32': 1}[mfcc_precision]
        vec = int(getattr(params, 'vectorizer', Vectorizer.mfccs))
        if vec == Vectorizer.mels:          # the mels entry is the mfccs pipeline without its DCT (offline form)
            vec = Vectorizer.mfccs
        p = PeParams(params.sample_rate, params.window_samples, params.hop_samples, params.n_fft,
                     params.n_filt, params.n_mfcc, params.n_features, int(bool(params.use_delta)), prec,
                     {'f32': 0, 'bf16': 1}[gru_precision], vec, {'f32': 0, 'bf16': 1}[ring_precision])
        if mel_filters is None:
            bank = speechpy_filterbank if vec == Vectorizer.speechpy_mfccs else mel_filterbank
            mel_filters = bank(params.sample_rate, params.n_filt, params.n_fft // 2 + 1)
        mel = np.ascontiguousarray(mel_filters, dtype=np.float64)
        if mel.shape != (params.n_filt, params.n_fft // 2 + 1):
            raise ValueError('mel filterbank has shape %r' % (mel.shape,))
        layers = weights['gru']
        keep = []                      # keep numpy buffers alive across the call
        arr = (PeGruLayer * len(layers))()
        for i, (k, rk, b) in enumerate(layers):
            k = np.ascontiguousarray(k, dtype=np.float32)
            rk = np.ascontiguousarray(rk, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            units = rk.shape[0]
            if k.shape[1] != 3 * units or rk.shape != (units, 3 * units) or b.shape != (3 * units,):
                raise ValueError('GRU layer %d has inconsistent shapes' % i)
            keep += [k, rk, b]
            arr[i] = PeGruLayer(k.shape[0], units, _fptr(k), _fptr(rk), _fptr(b))
        dk = np.ascontiguousarray(weights['dense_kernel'], dtype=np.float32).reshape(-1)
        db = float(np.asarray(weights['dense_bias'], dtype=np.float32).reshape(-1)[0])
        w = PeWeights(len(layers), arr, _fptr(dk), db)
        rc = self._lib.pe_create(C.byref(p), mel.ctypes.data_as(C.POINTER(C.c_double)), C.byref(w),
                                 self.n_streams, int(device), C.byref(self._h))
        if rc != PE_OK:
            msg = self._lib.pe_last_global_error().decode()
            self._h = C.c_void_p()
            self._raise(rc, msg)
        self.units = layers[-1][1].shape[0]
        self._win_hop = (int(params.window_samples), int(params.hop_samples))

    # -- errors -------------------------------------------------------------------------
    @staticmethod
    def _raise(rc, msg):
        if rc == PE_ERR_EOF:
            raise EOFError
        if rc == PE_ERR_INVALID:
            raise ValueError(msg)
        if rc == PE_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        if rc == PE_ERR_NOMEM:
            raise MemoryError(msg)
        raise EngineError(rc, msg)

    def _check(self, rc):
        if rc != PE_OK:
            self._raise(rc, self._lib.pe_last_error(self._h).decode())

    def _pcm(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype='<i2')
        if pcm.ndim == 1:
            pcm = pcm.reshape(1, -1)
        if pcm.ndim != 2 or pcm.shape[0] != self.n_streams:
            raise ValueError('pcm must be int16 [n_streams=%d, chunk_samples], got %r' %
                             (self.n_streams, pcm.shape))
        return pcm

    # -- host entry points --------------------------------------------------------------
    def update(self, pcm) -> np.ndarray:
        """int16 [n_streams, chunk] -> raw network outputs float32 [n_streams]."""
        pcm = self._pcm(pcm)
        out = np.empty(self.n_streams, dtype=np.float32)
        self._check(self._lib.pe_update(self._h, pcm.ctypes.data, pcm.shape[1], out.ctypes.data))
        return out

    def update_subset(self, stream_ids, pcm) -> np.ndarray:
        """The streams named in ``stream_ids`` (unique, in range) take one chunk each -- pcm int16 [len(stream_ids), chunk] --
        and every other stream stays as it is; -> raw outputs float32 [len(stream_ids)] in the order of ``stream_ids``."""
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32).reshape(-1)
        pcm = np.ascontiguousarray(pcm, dtype='<i2')
        if pcm.ndim == 1:
            pcm = pcm.reshape(1, -1)
        if pcm.ndim != 2 or pcm.shape[0] != ids.size:
            raise ValueError('pcm must be int16 [%d active streams, chunk_samples], got %r' % (ids.size, pcm.shape))
        out = np.empty(ids.size, dtype=np.float32)
        self._check(self._lib.pe_update_subset(self._h, ids.ctypes.data, ids.size, pcm.ctypes.data, pcm.shape[1], out.ctypes.data))
        return out

    def update_subset_device(self, ids_ptr: int, n_active: int, pcm_ptr: int, chunk_samples: int, out_ptr: int, stream: int = 0):
        self._check(self._lib.pe_update_subset_device(self._h, ids_ptr, n_active, pcm_ptr, chunk_samples, out_ptr, stream))

    def set_renumber_at(self, call_number: int):
        self._check(self._lib.pe_set_renumber_at(self._h, int(call_number)))

    # -- host-fed pipeline (pe_update_async / pe_wait): scripts/engine.py:60-63 hands over host bytes per chunk ----------
    def host_array(self, shape, dtype) -> np.ndarray:
        """A numpy array over pinned, device-visible host memory of this engine (pe_host_alloc): ``update_async`` reads PCM
        from / writes probabilities to such arrays without a staging copy.  The memory belongs to the engine (pe_destroy frees
        it), so the array keeps the engine alive: ``close()`` -- explicit or by garbage collection -- takes effect only once
        the last such array (and every view of it) is gone."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self._lib.pe_host_alloc(self._h, max(n, 1), C.byref(p)))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        # `buf` is the base object of `arr` and of every view of it; the finalizer holds the engine (a bound method) until buf dies
        self._views += 1
        weakref.finalize(buf, self._view_released).atexit = False      # (at interpreter exit the HIP runtime may already be gone)
        return arr

    def _view_released(self):
        self._views -= 1
        if self._views == 0 and self._close_pending:
            self.close()

    def update_async(self, pcm: np.ndarray, out: np.ndarray = None) -> np.ndarray:
        """Enqueue one update ([n_streams, chunk] int16) and return the float32 [n_streams] array its probabilities will be
        in after ``wait()`` (or once 3 more updates have been enqueued).  Up to 3 updates are in flight: the next chunk
        crosses PCIe while this one runs.  ``pcm`` is handed to ``hipMemcpyAsync`` as it is: PAGEABLE memory is staged by the
        HIP runtime before the call returns (the array is free again at once); memory the runtime knows as PINNED --
        ``host_array``, but also ``torch.Tensor.pin_memory()`` / hipHostRegister'ed buffers -- is read by the DMA engine after
        the call returns and must stay untouched until ``wait()`` or until 3 more updates have been enqueued."""
        pcm = self._pcm(pcm)
        if out is None:
            out = np.empty(self.n_streams, dtype=np.float32)
        if out.dtype != np.float32 or out.size != self.n_streams or not out.flags.c_contiguous:
            raise ValueError('out must be a contiguous float32 array of %d elements' % self.n_streams)
        self._check(self._lib.pe_update_async(self._h, pcm.ctypes.data, pcm.shape[1], out.ctypes.data))
        self._async_keep = (self._async_keep + [(pcm, out)])[-4:]          # the buffers of the updates in flight stay alive
        return out

    def wait(self):
        self._check(self._lib.pe_wait(self._h))
        self._async_keep = []

    def reserve_updates(self, max_updates: int, max_chunk_samples: int):
        """Size the engine for update_many (restarts all streams)."""
        self._check(self._lib.pe_reserve_updates(self._h, int(max_updates), int(max_chunk_samples)))

    def update_many(self, pcm) -> np.ndarray:
        """int16 [n_updates, n_streams, chunk] -> raw outputs float32 [n_updates, n_streams]; identical to
        n_updates consecutive update() calls."""
        pcm = np.ascontiguousarray(pcm, dtype='<i2')
        if pcm.ndim != 3 or pcm.shape[1] != self.n_streams:
            raise ValueError('pcm must be int16 [n_updates, n_streams=%d, chunk_samples], got %r' % (self.n_streams, pcm.shape))
        out = np.empty((pcm.shape[0], self.n_streams), dtype=np.float32)
        self._check(self._lib.pe_update_many(self._h, pcm.ctypes.data, pcm.shape[2], pcm.shape[0], out.ctypes.data))
        return out

    def update_many_device(self, pcm_ptr: int, chunk_samples: int, n_updates: int, out_ptr: int, stream: int = 0):
        self._check(self._lib.pe_update_many_device(self._h, pcm_ptr, chunk_samples, n_updates, out_ptr, stream))

    def update_vectors(self, pcm, want_features=True):
        pcm = self._pcm(pcm)
        feats = np.empty((self.n_streams, self.n_features, self.n_mfcc), dtype=np.float32) if want_features else None
        self._check(self._lib.pe_update_vectors(self._h, pcm.ctypes.data, pcm.shape[1],
                                                feats.ctypes.data if want_features else None))
        return feats

    def get_vectors(self) -> np.ndarray:
        """Current feature windows float32 [n_streams, T, F], oldest row first."""
        feats = np.empty((self.n_streams, self.n_features, self.n_mfcc), dtype=np.float32)
        self._check(self._lib.pe_get_vectors(self._h, feats.ctypes.data))
        return feats

    def set_vectors(self, feats):
        """Restart every stream with the given feature windows [n_streams, T, F] already emitted."""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        if feats.shape != (self.n_streams, self.n_features, self.n_mfcc):
            raise ValueError('expected [%d, %d, %d] features, got %r' % (self.n_streams, self.n_features, self.n_mfcc, feats.shape))
        self._check(self._lib.pe_set_vectors(self._h, feats.ctypes.data))

    def predict(self, feats) -> np.ndarray:
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        if feats.ndim != 3 or feats.shape[1:] != (self.n_features, self.feature_size):
            raise ValueError('inputs must be [N, %d, %d], got %r' % (self.n_features, self.feature_size, feats.shape))
        out = np.empty((feats.shape[0], 1), dtype=np.float32)
        self._check(self._lib.pe_predict(self._h, feats.ctypes.data, feats.shape[0], out.ctypes.data))
        return out

    def vectorize_raw(self, audio) -> np.ndarray:
        """float64 audio [n] -> MFCC frames float64 [1 + (n - window)//hop, n_mfcc] (stateless)."""
        audio = np.ascontiguousarray(audio, dtype=np.float64).reshape(-1)
        win, hop = self._win_hop
        max_frames = 1 + (audio.size - win) // hop if audio.size >= win else 0
        out = np.empty((max_frames, self.n_mfcc), dtype=np.float64)
        n = C.c_int64(0)
        self._check(self._lib.pe_vectorize_raw(self._h, audio.ctypes.data if audio.size else None, audio.size,
                                               out.ctypes.data if max_frames else None, max_frames, C.byref(n)))
        return out[:n.value]

    def vectorize_mels(self, audio) -> np.ndarray:
        """float64 audio [n] -> log-mel frames float64 [1 + (n - window)//hop, n_filt] (stateless; Vectorizer.mels)."""
        audio = np.ascontiguousarray(audio, dtype=np.float64).reshape(-1)
        win, hop = self._win_hop
        max_frames = 1 + (audio.size - win) // hop if audio.size >= win else 0
        out = np.empty((max_frames, self.n_filt), dtype=np.float64)
        n = C.c_int64(0)
        self._check(self._lib.pe_vectorize_mels(self._h, audio.ctypes.data if audio.size else None, audio.size,
                                                out.ctypes.data if max_frames else None, max_frames, C.byref(n)))
        return out[:n.value]

    def evaluate(self, audio, hop_frames: int) -> np.ndarray:
        """Whole recording (float64 audio) -> raw outputs [n_windows, 1] of the windows ending at frames
        range(T, n_frames, hop_frames): simulate.py:92-104 in one call."""
        audio = np.ascontiguousarray(audio, dtype=np.float64).reshape(-1)
        win, hop = self._win_hop
        n_frames = 1 + (audio.size - win) // hop if audio.size >= win else 0
        n_win = max(0, -(-(n_frames - self.n_features) // int(hop_frames))) if n_frames > self.n_features else 0
        out = np.empty((n_win, 1), dtype=np.float32)
        n = C.c_int64(0)
        self._check(self._lib.pe_evaluate(self._h, audio.ctypes.data if audio.size else None, audio.size,
                                          int(hop_frames), out.ctypes.data if n_win else None, n_win, C.byref(n)))
        return out[:n.value]

    def set_decoder(self, decoder):
        """Upload a ThresholdDecoder (its cumulative table and scalars) for pe_decode*."""
        cd = np.ascontiguousarray(decoder.cd, dtype=np.float64)
        self._check(self._lib.pe_set_decoder(self._h, cd.ctypes.data if cd.size else None, cd.size,
                                             int(decoder.min_out), int(decoder.out_range), float(decoder.center)))

    def set_trigger(self, chunk_size: int, sensitivity: float = 0.5, trigger_level: int = 3):
        self._check(self._lib.pe_set_trigger(self._h, int(chunk_size), float(sensitivity), int(trigger_level)))

    def decode(self, raw, want_fired=False):
        """raw float32 [n_streams] -> decoded confidences float64 [n_streams] (, fired bool [n_streams])."""
        raw = np.ascontiguousarray(raw, dtype=np.float32).reshape(-1)
        if raw.size != self.n_streams:
            raise ValueError('expected %d raw outputs' % self.n_streams)
        conf = np.empty(self.n_streams, dtype=np.float64)
        fired = np.zeros(self.n_streams, dtype=np.uint8)
        self._check(self._lib.pe_decode(self._h, raw.ctypes.data, conf.ctypes.data, fired.ctypes.data))
        return (conf, fired.astype(bool)) if want_fired else conf

    def clear(self, mask=None):
        if mask is None:
            self._check(self._lib.pe_clear(self._h, None))
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            if m.shape != (self.n_streams,):
                raise ValueError('mask must have shape (%d,)' % self.n_streams)
            self._check(self._lib.pe_clear(self._h, m.ctypes.data))

    # -- device entry points (pointers are ints, e.g. torch.Tensor.data_ptr()) ---------------
    def update_device(self, pcm_ptr: int, chunk_samples: int, out_ptr: int, stream: int = 0, keep: bool = False):
        """keep=True: the caller promises the chunks at pcm_ptr stay alive and unchanged until the next update's work is
        done -- the leftover samples then stay there instead of being copied to the engine's carry (pe_update_device_keep)."""
        fn = self._lib.pe_update_device_keep if keep else self._lib.pe_update_device
        self._check(fn(self._h, pcm_ptr, chunk_samples, out_ptr, stream))

    def update_vectors_device(self, pcm_ptr: int, chunk_samples: int, feats_ptr: int = 0, stream: int = 0):
        self._check(self._lib.pe_update_vectors_device(self._h, pcm_ptr, chunk_samples, feats_ptr or None, stream))

    def run_device(self, out_ptr: int, stream: int = 0):
        self._check(self._lib.pe_run_device(self._h, out_ptr, stream))

    def predict_device(self, feats_ptr: int, n: int, out_ptr: int, stream: int = 0):
        self._check(self._lib.pe_predict_device(self._h, feats_ptr, n, out_ptr, stream))

    # -- introspection ------------------------------------------------------------------
    def info(self) -> PeInfo:
        i = PeInfo()
        self._check(self._lib.pe_get_info(self._h, C.byref(i)))
        return i

    def stream_state(self):
        q = np.empty(self.n_streams, dtype=np.int32)
        kc = np.empty(self.n_streams, dtype=np.uint32)
        ke = np.empty(self.n_streams, dtype=np.uint32)
        self._check(self._lib.pe_get_stream_state(self._h, q.ctypes.data, kc.ctypes.data, ke.ctypes.data))
        return q, kc, ke

    def set_fused(self, enabled: bool):
        self._check(self._lib.pe_set_fused(self._h, int(bool(enabled))))

    def set_input_projection(self, enabled: bool):
        """Store x.W + b per frame beside the feature ring (True) or recompute it in the network (False); restarts the streams."""
        self._check(self._lib.pe_set_input_projection(self._h, int(bool(enabled))))

    def set_gru_waves(self, waves: int):
        self._check(self._lib.pe_set_gru_waves(self._h, int(waves)))

    def set_gru_tiling(self, tiling: int):
        """-1 automatic, 0 classic four-tile layout, 1 re-tiled stock width (csrc/gru_cw_device.h), 2 float32 products on the
        bf16 matrix pipe (csrc/gru_x3_device.h; automatic above four stream tiles per compute unit).  bf16 networks: 1 / -1 = five
        gate values per lane where the network fits (csrc/gru_b20_device.h), 0 = eight (csrc/gru_bf16_device.h)."""
        self._check(self._lib.pe_set_gru_tiling(self._h, int(tiling)))

    def gru_tiling(self) -> int:
        """The form this engine's network launches take now (pe_get_gru_tiling): float32 networks of <= 32 units 0 / 1 / 2 as
        above; wide / stacked networks 0 (f32-input MFMAs) or 2 (float32 products on the bf16 pipe); bf16-operand networks
        1 (five gate values per lane) or 0 (eight)."""
        return int(self._lib.pe_get_gru_tiling(self._h))

    def set_timing(self, enabled: bool):
        self._check(self._lib.pe_set_timing(self._h, int(bool(enabled))))

    def last_timing(self):
        a, b = C.c_float(0), C.c_float(0)
        self._check(self._lib.pe_get_last_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        """pe_destroy.  While arrays from ``host_array`` are still referenced the destruction is deferred until the last one
        is gone (their memory is the engine's): nothing can read freed pinned memory through a stale array."""
        if getattr(self, '_h', None) and self._h.value:
            if getattr(self, '_views', 0) > 0:
                self._close_pending = True
                try:
                    self._lib.pe_wait(self._h)       # nothing of this engine stays in flight behind a "closed" handle
                except Exception:
                    pass
                return
            self._lib.pe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
