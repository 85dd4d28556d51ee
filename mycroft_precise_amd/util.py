"""PCM helpers: drop-in for the hot-path part of ``precise.util`` (/root/reference/precise/util.py:25-42)."""
import numpy as np


class InvalidAudio(ValueError):
    """Audio is not in the expected format (util.py:25-27)."""


def buffer_to_audio(buffer: bytes) -> np.ndarray:
    """Raw little-endian int16 mono bytes -> float32 in [-1, 1): value / 32768.0 (util.py:35-37)."""
    return np.frombuffer(buffer, dtype='<i2').astype(np.float32, order='C') / 32768.0


def audio_to_buffer(audio: np.ndarray) -> bytes:
    """Inverse of buffer_to_audio (util.py:40-42)."""
    return (np.asarray(audio) * 32768).astype('<i2').tobytes()


def chunk_audio(audio: np.ndarray, chunk_size: int):
    """Whole chunks of ``chunk_size`` samples, all but the final one (util.py:30-32)."""
    for end in range(chunk_size, len(audio), chunk_size):
        yield audio[end - chunk_size:end]


def pcm16_from(stream_input) -> np.ndarray:
    """Whatever Listener.update accepts as audio -> contiguous int16 samples for the device.

    bytes -> reinterpreted as '<i2' (ValueError on an odd byte count, like np.fromstring);
    int16 ndarray -> as is;  float ndarray -> int16 when every sample is exactly k/32768 (what
    buffer_to_audio produces); any other float audio cannot be fed to the int16 PCM path.
    """
    if isinstance(stream_input, (bytes, bytearray, memoryview)):
        if len(stream_input) % 2:
            raise ValueError('string size must be a multiple of element size')
        return np.frombuffer(stream_input, dtype='<i2')
    a = np.asarray(stream_input)
    if a.dtype == np.int16:
        return np.ascontiguousarray(a.reshape(-1))
    if a.dtype.kind == 'f':
        scaled = a.astype(np.float64).reshape(-1) * 32768.0
        ints = np.rint(scaled)
        if np.all(ints == scaled) and (ints.size == 0 or (ints.min() >= -32768 and ints.max() <= 32767)):
            return ints.astype('<i2')
        raise TypeError('the HIP listener consumes int16 PCM: float audio must hold exact k/32768 '
                        'sample values (as produced by buffer_to_audio)')
    raise TypeError('unsupported audio dtype: %s' % a.dtype)
