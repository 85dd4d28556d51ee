"""
CPU restatement of sonopy 0.1.2 (third-party dependency of the reference, pinned at
``/root/reference/requirements.txt:35``; call sites
``/root/reference/precise/vectorization.py:24,31-39``).  TEST INFRASTRUCTURE ONLY.

sonopy's source is NOT under /root/reference and cannot be installed here (no network),
so this file restates its published algorithm from the package's documented behaviour.
**Parity with the real sonopy bits is UNPINNED**; every quirk that matters numerically is
named below so that it can be diffed against the real source when it is obtainable.

Named quirks (all float64, numpy):
  Q1  framing: frame k = audio[k*hop : k*hop + window], one frame per full window,
      n = 1 + (len - window) // hop, NO padding, NO pre-emphasis, NO window function.
  Q2  ``np.fft.rfft(frames, n=fft_size)``: with window(1600) > fft_size(512) numpy CROPS
      every frame to its first 512 samples -- the last 1088 samples of a window never
      reach the FFT.
  Q3  power = (re^2 + im^2) / fft_size           (fft_size//2 + 1 = 257 bins)
  Q4  mel filterbank: num_filt+2 grid points equally spaced in mels between
      mel(0) and mel(sample_rate) (sample_rate, NOT Nyquist), mel(f)=1127 ln(1+f/700),
      mapped to bin indices ``int(hz * n_bins / sample_rate)``, duplicates pushed forward
      (``DUPLICATES = 'push'``; ``'keep'`` = left alone, the other possible behaviour of the real
      package's ``correct_grid`` -- the two differ only for non-stock settings with colliding points);
      filter i rises linearly 0->1 over [left, mid) and falls 1->0 over [mid, right)
      (``linspace(..., endpoint=False)``, so the peak 1.0 sits at bin ``mid`` and bin
      ``left`` has weight 0).
  Q5  safe_log(x) = log(clip(x, eps_float64, None)), eps = 2**-52.
  Q6  mfcc = DCT-II (scipy.fftpack.dct, norm='ortho') of the log-mel energies, first
      num_coeffs columns.
  Q7  coefficient 0 is REPLACED by safe_log(sum over the 257 power bins).
"""
import numpy as np
from scipy.fftpack import dct

EPS = np.finfo(float).eps  # 2**-52, Q5


def frame_starts(n_samples: int, window: int, hop: int):
    """Q1: start offsets of every full window."""
    if n_samples < window:
        return np.empty((0,), dtype=np.int64)
    return np.arange(0, n_samples - window + 1, hop, dtype=np.int64)


def power_spec(audio, window_stride=(160, 80), fft_size=512):
    """Q1-Q3.  audio: 1-D float array -> [n_frames, fft_size//2+1] float64."""
    audio = np.asarray(audio, dtype=np.float64)
    window, hop = window_stride
    starts = frame_starts(len(audio), window, hop)
    if len(starts) == 0:
        spec = np.fft.rfft(np.empty((0, window)), n=fft_size)
    else:
        idx = starts[:, None] + np.arange(window)[None, :]
        spec = np.fft.rfft(audio[idx], n=fft_size)   # Q2: crops to fft_size samples
    return (spec.real ** 2 + spec.imag ** 2) / fft_size


def _hz_to_mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def _mel_to_hz(m):
    return 700.0 * (np.exp(m / 1127.0) - 1.0)


def _push_duplicates(idx):
    """Q4: make the grid strictly increasing by pushing repeated points forward."""
    out = []
    offset = 0
    prev = idx[0] - 1
    for i in idx:
        offset = max(0, offset + prev + 1 - i)
        out.append(i + offset)
        prev = i
    return out


# Q4 fork (see the header): 'push' = correct_grid fires and repeated grid points move forward, 'keep' = it never fires
# (ndarray argument: ``[x[0] - 1] + x`` broadcasts).  Identical for the stock 20 filters.  Module default, like the product's
# mycroft_precise_amd.vectorization.sonopy_duplicates.
DUPLICATES = 'push'


def filterbanks(sample_rate, num_filt, fft_len, duplicates=None):
    """Q4.  -> [num_filt, fft_len] float64 triangular filters."""
    mode = DUPLICATES if duplicates is None else duplicates
    if mode not in ('push', 'keep'):
        raise ValueError("duplicates must be 'push' or 'keep'")
    grid_mels = np.linspace(_hz_to_mel(0.0), _hz_to_mel(float(sample_rate)), num_filt + 2, True)
    grid_hz = _mel_to_hz(grid_mels)
    grid_idx = [int(v) for v in (grid_hz * fft_len / sample_rate).astype(int)]
    if mode == 'push':
        grid_idx = _push_duplicates(grid_idx)
    banks = np.zeros((num_filt, fft_len))
    for i in range(num_filt):
        left, mid, right = grid_idx[i], grid_idx[i + 1], grid_idx[i + 2]
        banks[i, left:mid] = np.linspace(0.0, 1.0, mid - left, False)
        banks[i, mid:right] = np.linspace(1.0, 0.0, right - mid, False)
    return banks


def safe_log(x):
    """Q5."""
    return np.log(np.clip(x, EPS, None))


def mel_spec(audio, sample_rate, window_stride=(160, 80), fft_size=512, num_filt=20):
    spec = power_spec(audio, window_stride, fft_size)
    return safe_log(np.dot(spec, filterbanks(sample_rate, num_filt, spec.shape[1]).T))


def mfcc_spec(audio, sample_rate, window_stride=(160, 80), fft_size=512, num_filt=20,
              num_coeffs=13, return_parts=False):
    """Q1-Q7.  -> [n_frames, min(num_filt, num_coeffs)] float64."""
    powers = power_spec(audio, window_stride, fft_size)
    if powers.size == 0:
        return np.empty((0, min(num_filt, num_coeffs)))
    filters = filterbanks(sample_rate, num_filt, powers.shape[1])
    mels = safe_log(np.dot(powers, filters.T))
    mfccs = dct(mels, norm='ortho')[:, :num_coeffs]
    mfccs[:, 0] = safe_log(np.sum(powers, 1))      # Q7
    if return_parts:
        return powers, filters, mels, mfccs
    return mfccs


def mfcc_from_frames(frames512, sample_rate=16000, fft_size=512, num_filt=20, num_coeffs=13):
    """Batched form used by the fast oracle: frames512 [..., fft_size] float64 that are
    ALREADY the cropped first fft_size samples of each window (Q2).  Same arithmetic as
    mfcc_spec from Q3 on."""
    spec = np.fft.rfft(np.asarray(frames512, dtype=np.float64), n=fft_size)
    powers = (spec.real ** 2 + spec.imag ** 2) / fft_size
    filters = filterbanks(sample_rate, num_filt, powers.shape[-1])
    mels = safe_log(np.dot(powers, filters.T))
    mfccs = dct(mels, norm='ortho')[..., :num_coeffs]
    mfccs[..., 0] = safe_log(np.sum(powers, -1))
    return mfccs
