"""
CPU restatement of speechpy-fast 2.4 (third-party dependency of the reference, pinned at
``/root/reference/requirements.txt:36``), the LEGACY vectorizer the reference selects for every
``<model>.params`` file without a ``vectorizer`` key (``/root/reference/precise/params.py:147,155``); call site
``/root/reference/precise/vectorization.py:40-42``:

    speechpy.feature.mfcc(x, pr.sample_rate, pr.window_t, pr.hop_t, pr.n_mfcc, pr.n_filt, pr.n_fft)

TEST INFRASTRUCTURE ONLY.  speechpy's source is NOT under /root/reference and cannot be installed here (no
network), so this file restates its published algorithm (speechpy 2.x ``feature.mfcc`` / ``feature.mfe`` /
``feature.filterbanks`` / ``processing.stack_frames`` / ``processing.power_spectrum`` / ``functions.*``) from the
package's documented behaviour.  **Parity with the real speechpy bits is UNPINNED**; every quirk that matters
numerically is named so that it can be diffed against the real source when it is obtainable.

Named quirks (all float64, numpy):
  S1  framing (``stack_frames(..., zero_padding=False)``): frame length ``int(round(fs * frame_length))``,
      stride ``round(fs * frame_stride)``; the NUMBER OF FRAMES is ``floor((len - frame_len) / stride)`` -- one
      fewer than the count of full windows (a signal of exactly one window yields NO frame); frame k =
      signal[k*stride : k*stride + frame_len]; rectangular window (``filter = ones``).
  S2  ``power_spectrum`` = ``|rfft(frames, n=fft_length)|**2 / fft_length``: as in sonopy, numpy CROPS every
      1600-sample frame to its first 512 samples; ``np.absolute`` then ``np.square`` (a square root and a
      square, not re*re + im*im: last-bit differences from sonopy's power).
  S3  frame energy = sum of the 257 power bins; ``zero_handling``: exact zeros become float64 eps
      (``np.where(x == 0, eps, x)`` -- NOT a clip: 0 < x < eps stays x).
  S4  filterbank: ``num_filters + 2`` points equally spaced in mels, mel(f) = 1127 ln(1 + f/700), between
      ``low_freq or 300`` -- the caller's 0 Hz is falsy, so the bank STARTS AT 300 Hz -- and fs/2;
      bin = ``floor((coefficients + 1) * hz / fs)`` with coefficients = 257 (so the top filter ends at bin 129,
      i.e. 4 kHz: speechpy's well-known bin-scale slip); filter i = triangle over [left, right] with peak 1.0
      at ``middle``, zero AT ``left`` and ``right``.
  S5  features = power . filterbank^T, ``zero_handling``, ``np.log``.
  S6  DCT-II (scipy.fftpack.dct, norm='ortho') of the log energies, first ``num_cepstral`` columns.
  S7  ``dc_elimination=True`` (default): coefficient 0 is REPLACED by log(frame energy).
"""
import math
import types

import numpy as np
from scipy.fftpack import dct

EPS = np.finfo(float).eps


def zero_handling(x):                                           # S3
    return np.where(x == 0, EPS, x)


def frequency_to_mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_to_frequency(mel):
    return 700.0 * (np.exp(mel / 1127.0) - 1.0)


def triangle(x, left, middle, right):                           # S4
    out = np.zeros(x.shape)
    first_half = np.logical_and(left < x, x <= middle)
    out[first_half] = (x[first_half] - left) / (middle - left)
    second_half = np.logical_and(middle <= x, x < right)
    out[second_half] = (right - x[second_half]) / (right - middle)
    return out


def frame_geometry(sampling_frequency, frame_length, frame_stride):
    return int(np.round(sampling_frequency * frame_length)), int(np.round(sampling_frequency * frame_stride))


def n_frames(n_samples, frame_len, stride):
    """S1: floor((len - frame_len) / stride), never negative."""
    return max(0, int(math.floor((n_samples - frame_len) / float(stride))))


def stack_frames(sig, sampling_frequency, frame_length=0.020, frame_stride=0.020, zero_padding=False):
    """S1 (the ``zero_padding=False`` branch is the only one ``mfe`` uses)."""
    assert sig.ndim == 1
    assert not zero_padding
    frame_len, stride = frame_geometry(sampling_frequency, frame_length, frame_stride)
    n = n_frames(sig.shape[0], frame_len, stride)
    idx = np.arange(frame_len)[None, :] + (np.arange(n) * stride)[:, None]
    return sig[idx.astype(np.int64)] * np.ones((frame_len,))[None, :]


def power_spectrum(frames, fft_points=512):                     # S2
    return 1.0 / fft_points * np.square(np.absolute(np.fft.rfft(frames, n=fft_points, axis=-1, norm=None)))


def filterbanks(num_filter, coefficients, sampling_freq, low_freq=None, high_freq=None):
    """S4.  -> [num_filter, coefficients] float64."""
    high_freq = high_freq or sampling_freq / 2
    low_freq = low_freq or 300
    assert high_freq <= sampling_freq / 2
    assert low_freq >= 0
    mels = np.linspace(frequency_to_mel(low_freq), frequency_to_mel(high_freq), num_filter + 2)
    hertz = mel_to_frequency(mels)
    freq_index = (np.floor((coefficients + 1) * hertz / sampling_freq)).astype(int)
    bank = np.zeros([num_filter, coefficients])
    for i in range(num_filter):
        left, middle, right = int(freq_index[i]), int(freq_index[i + 1]), int(freq_index[i + 2])
        z = np.linspace(left, right, num=right - left + 1)
        bank[i, left:right + 1] = triangle(z, left=left, middle=middle, right=right)
    return bank


def mfe(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_filters=40, fft_length=512,
        low_frequency=0, high_frequency=None):
    signal = signal.astype(float)
    frames = stack_frames(signal, sampling_frequency, frame_length, frame_stride, zero_padding=False)
    high_frequency = high_frequency or sampling_frequency / 2
    power = power_spectrum(frames, fft_length)
    coefficients = power.shape[1]
    frame_energies = zero_handling(np.sum(power, 1))            # S3
    bank = filterbanks(num_filters, coefficients, sampling_frequency, low_frequency, high_frequency)
    features = zero_handling(np.dot(power, bank.T))             # S5
    return features, frame_energies


def mfcc(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_cepstral=13, num_filters=40,
         fft_length=512, low_frequency=0, high_frequency=None, dc_elimination=True):
    """S1-S7.  -> [n_frames, num_cepstral] float64."""
    feature, energy = mfe(signal, sampling_frequency, frame_length, frame_stride, num_filters, fft_length,
                          low_frequency, high_frequency)
    if len(feature) == 0:
        return np.empty((0, num_cepstral))
    feature = np.log(feature)
    feature = dct(feature, type=2, axis=-1, norm='ortho')[:, :num_cepstral]
    if dc_elimination:
        feature[:, 0] = np.log(energy)                          # S7
    return feature


def mfcc_from_frames(frames512, sample_rate=16000, fft_size=512, num_filt=20, num_coeffs=13):
    """Batched form for the fast oracle: frames512 [..., fft_size] are ALREADY the cropped first fft_size samples
    of each frame (S2).  Same arithmetic as ``mfcc`` from S2 on."""
    power = power_spectrum(np.asarray(frames512, dtype=np.float64), fft_size)
    energy = zero_handling(np.sum(power, -1))
    bank = filterbanks(num_filt, power.shape[-1], sample_rate, 0, None)
    feats = np.log(zero_handling(np.dot(power, bank.T)))
    out = dct(feats, type=2, axis=-1, norm='ortho')[..., :num_coeffs]
    out[..., 0] = np.log(energy)
    return out


# ``speechpy.feature.mfcc`` is how the reference reaches it (vectorization.py:40)
feature = types.SimpleNamespace(mfcc=mfcc, mfe=mfe, filterbanks=filterbanks)
processing = types.SimpleNamespace(stack_frames=stack_frames, power_spectrum=power_spectrum)
functions = types.SimpleNamespace(zero_handling=zero_handling, triangle=triangle,
                                  frequency_to_mel=frequency_to_mel, mel_to_frequency=mel_to_frequency)
