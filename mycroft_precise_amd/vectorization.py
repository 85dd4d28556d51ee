"""
Audio -> feature vectors: drop-in for ``precise.vectorization``
(/root/reference/precise/vectorization.py:31-89).  The ``vectorizers`` dict is the reference's
plug-in seam for the front end; its ``Vectorizer.mfccs`` and ``Vectorizer.mels`` entries are served by
the HIP kernels (stateless whole-buffer form, ``pe_vectorize_raw`` / ``pe_vectorize_mels``).  There is
no CPU implementation here.
"""
import numpy as np

from .params import pr, Vectorizer
from .util import InvalidAudio


def mel_filterbank(sample_rate: int, num_filt: int, n_bins: int) -> np.ndarray:
    """
    Triangular mel filters [num_filt, n_bins] (float64), the constant table handed to
    ``pe_create``.  Construction follows the filterbank of the vectorizer the reference calls
    (vectorization.py:36-39 -> sonopy 0.1.2): num_filt+2 points equally spaced on the mel scale
    m(f) = 1127 ln(1 + f/700) between 0 Hz and ``sample_rate`` Hz (sic), mapped to bins with
    int(hz * n_bins / sample_rate), repeated points pushed forward, each filter rising over
    [left, mid) and falling over [mid, right) with endpoint-free linspaces.
    """
    top = 1127.0 * np.log(1.0 + float(sample_rate) / 700.0)
    mels = np.linspace(0.0, top, num_filt + 2, True)
    hz = 700.0 * (np.exp(mels / 1127.0) - 1.0)
    raw = (hz * n_bins / sample_rate).astype(int).tolist()
    pts, shift, last = [], 0, raw[0] - 1
    for v in raw:
        shift = max(0, shift + last + 1 - v)
        pts.append(v + shift)
        last = v
    bank = np.zeros((num_filt, n_bins), dtype=np.float64)
    for f in range(num_filt):
        lo, mid, hi = pts[f], pts[f + 1], pts[f + 2]
        bank[f, lo:mid] = np.linspace(0.0, 1.0, mid - lo, False)
        bank[f, mid:hi] = np.linspace(1.0, 0.0, hi - mid, False)
    return bank


_offline = {}


def _offline_engine():
    """One stateless engine per parameter set, created on first use."""
    from ._lib import HipEngine
    key = (pr.sample_rate, pr.window_samples, pr.hop_samples, pr.n_fft, pr.n_filt, pr.n_mfcc)
    eng = _offline.get(key)
    if eng is None:
        snap = pr.copy()
        snap.__dict__['use_delta'] = False
        f = snap.n_mfcc
        dummy = {'gru': [(np.zeros((f, 3), np.float32), np.zeros((1, 3), np.float32), np.zeros(3, np.float32))],
                 'dense_kernel': np.zeros((1, 1), np.float32), 'dense_bias': np.zeros(1, np.float32)}
        eng = _offline[key] = HipEngine(snap, dummy, n_streams=1)
    return eng


def _mfccs_hip(audio: np.ndarray) -> np.ndarray:
    return _offline_engine().vectorize_raw(audio)


def _mels_hip(audio: np.ndarray) -> np.ndarray:
    """vectorization.py:32-35: log mel-filterbank energies [n, n_filt] -- the MFCC pipeline without its DCT."""
    return _offline_engine().vectorize_mels(audio)


def _no_kernel(name):
    def fn(audio):
        raise NotImplementedError('Vectorizer.%s has no HIP kernel (Vectorizer.mfccs and Vectorizer.mels do)' % name)
    return fn


# audio frames -> vectors (vectorization.py:31-43)
vectorizers = {
    Vectorizer.mels: _mels_hip,
    Vectorizer.mfccs: _mfccs_hip,
    Vectorizer.speechpy_mfccs: _no_kernel('speechpy_mfccs'),
}


def vectorize_raw(audio: np.ndarray) -> np.ndarray:
    """Feature vectors of a whole buffer, no length clipping (vectorization.py:46-50)."""
    if len(audio) == 0:
        raise InvalidAudio('Cannot vectorize empty audio!')
    return vectorizers[pr.vectorizer](audio)


def add_deltas(features: np.ndarray) -> np.ndarray:
    """Append first differences along time; the first row's delta is zero (vectorization.py:53-59)."""
    deltas = np.zeros_like(features)
    deltas[1:] = features[1:] - features[:-1]
    return np.concatenate([features, deltas], -1)


def vectorize(audio: np.ndarray) -> np.ndarray:
    """Last ``max_samples`` of audio -> exactly ``n_features`` rows: left zero pad or keep the tail
    (vectorization.py:62-84)."""
    if len(audio) > pr.max_samples:
        audio = audio[-pr.max_samples:]
    feats = vectorize_raw(audio)
    missing = pr.n_features - len(feats)
    if missing > 0:
        feats = np.concatenate([np.zeros((missing, feats.shape[1])), feats])
    elif missing < 0:
        feats = feats[-pr.n_features:]
    return feats


def vectorize_delta(audio: np.ndarray) -> np.ndarray:
    """vectorize + deltas (vectorization.py:87-89)."""
    return add_deltas(vectorize(audio))
