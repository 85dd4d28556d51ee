"""
Audio -> feature vectors: drop-in for ``precise.vectorization``
(/root/reference/precise/vectorization.py:31-89).  The ``vectorizers`` dict is the reference's
plug-in seam for the front end; all three entries (``mfccs``, ``mels``, and the legacy ``speechpy_mfccs`` that
old ``.params`` files select) are served by the HIP kernels (stateless whole-buffer form, ``pe_vectorize_raw`` /
``pe_vectorize_mels``).  There is no CPU implementation here.
"""
import warnings

import numpy as np

from .params import pr, Vectorizer
from .util import InvalidAudio


class UnverifiedFilterbank(UserWarning):
    """The requested (n_filt, n_fft) makes mel grid points collide: a case the restated sonopy algorithm could not
    be checked on (see ``mel_filterbank``)."""


# How repeated mel grid points are treated when a filterbank is built and the caller does not say: 'push' (what sonopy
# 0.1.2's ``correct_grid`` is written to do) or 'keep' (what it does if the correction never fires, see ``mel_filterbank``).
# A maintainer who can run the real package sets this once (or passes ``duplicates=`` / ``mel_filters=`` explicitly).
sonopy_duplicates = 'push'


def mel_filterbank(sample_rate: int, num_filt: int, n_bins: int, duplicates: str = None) -> np.ndarray:
    """
    Triangular mel filters [num_filt, n_bins] (float64), the constant table handed to
    ``pe_create``.  Construction follows the filterbank of the vectorizer the reference calls
    (vectorization.py:36-39 -> sonopy 0.1.2): num_filt+2 points equally spaced on the mel scale
    m(f) = 1127 ln(1 + f/700) between 0 Hz and ``sample_rate`` Hz (sic), mapped to bins with
    int(hz * n_bins / sample_rate), each filter rising over [left, mid) and falling over [mid, right)
    with endpoint-free linspaces.

    duplicates: what happens to REPEATED grid points (they exist only for non-stock settings, e.g. 40 filters at
    n_fft = 512; the stock 20 filters have none, and both choices then build the same table):
      'push' -- repeated points are pushed forward until the grid is strictly increasing: what sonopy's
                ``correct_grid`` is written to do;
      'keep' -- the grid is used as computed: what sonopy does if ``correct_grid`` receives an ndarray, for which its
                ``[x[0] - 1] + x`` broadcasts and the correction never fires.  A filter whose three points coincide is
                empty (its energy is 0, its log-mel value log(eps)).
    None = the module default ``sonopy_duplicates`` ('push').  Which of the two the real package does could not be
    checked offline (requirements.txt:35 pins sonopy==0.1.2; it is not installed here), hence the warning.
    """
    mode = sonopy_duplicates if duplicates is None else duplicates
    if mode not in ('push', 'keep'):
        raise ValueError("duplicates must be 'push' or 'keep', got %r" % (mode,))
    top = 1127.0 * np.log(1.0 + float(sample_rate) / 700.0)
    mels = np.linspace(0.0, top, num_filt + 2, True)
    hz = 700.0 * (np.exp(mels / 1127.0) - 1.0)
    raw = (hz * n_bins / sample_rate).astype(int).tolist()
    if len(set(raw)) != len(raw):
        warnings.warn("mel grid of %d filters over %d bins has repeated points: built with duplicates=%r; the filterbank for "
                      "this setting is UNVERIFIED against sonopy 0.1.2 (whether its correct_grid fires could not be checked) -- "
                      "switch with mel_filterbank(..., duplicates='push'|'keep') or vectorization.sonopy_duplicates"
                      % (num_filt, n_bins, mode), UnverifiedFilterbank, stacklevel=2)
    if mode == 'keep':
        pts = raw
    else:
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


def speechpy_filterbank(sample_rate: int, num_filt: int, n_bins: int) -> np.ndarray:
    """
    The filterbank of the legacy vectorizer (vectorization.py:40-42 -> speechpy-fast 2.4 ``feature.filterbanks``
    as ``feature.mfe`` calls it with low_frequency=0, high_frequency=None): num_filt+2 points equally spaced on
    the mel scale m(f) = 1127 ln(1 + f/700) between 300 Hz -- that library reads a lower edge of 0 as "unset" --
    and sample_rate/2, mapped to bins with floor((n_bins + 1) * hz / sample_rate); filter i is the triangle
    over [left, right] that is zero AT both ends and 1.0 at ``middle``.
    """
    lo = 1127.0 * np.log(1.0 + 300.0 / 700.0)
    hi = 1127.0 * np.log(1.0 + (sample_rate / 2) / 700.0)
    hz = 700.0 * (np.exp(np.linspace(lo, hi, num_filt + 2) / 1127.0) - 1.0)
    pts = np.floor((n_bins + 1) * hz / sample_rate).astype(int)
    bank = np.zeros((num_filt, n_bins), dtype=np.float64)
    for f in range(num_filt):
        left, mid, right = int(pts[f]), int(pts[f + 1]), int(pts[f + 2])
        x = np.linspace(left, right, num=right - left + 1)
        tri = np.zeros(x.shape)
        up = np.logical_and(left < x, x <= mid)
        tri[up] = (x[up] - left) / (mid - left)
        down = np.logical_and(mid <= x, x < right)
        tri[down] = (right - x[down]) / (right - mid)
        bank[f, left:right + 1] = tri
    return bank


_offline = {}


def _offline_engine(vectorizer=Vectorizer.mfccs):
    """One stateless engine per parameter set, created on first use."""
    from ._lib import HipEngine
    key = (pr.sample_rate, pr.window_samples, pr.hop_samples, pr.n_fft, pr.n_filt, pr.n_mfcc, int(vectorizer))
    eng = _offline.get(key)
    if eng is None:
        snap = pr.copy()
        snap.__dict__['use_delta'] = False
        snap.__dict__['vectorizer'] = vectorizer
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


def _speechpy_hip(audio: np.ndarray) -> np.ndarray:
    """vectorization.py:40-42: the legacy front end -- its own filterbank, one frame fewer per buffer, exact zeros
    (only) replaced by eps before the logarithms; same kernels."""
    return _offline_engine(Vectorizer.speechpy_mfccs).vectorize_raw(audio)


# audio frames -> vectors (vectorization.py:31-43)
vectorizers = {
    Vectorizer.mels: _mels_hip,
    Vectorizer.mfccs: _mfccs_hip,
    Vectorizer.speechpy_mfccs: _speechpy_hip,
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
