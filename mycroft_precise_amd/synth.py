"""
Synthetic workload definition: seeded 16 kHz int16 PCM streams and
random-init networks of the reference's topology (``/root/reference/precise/model.py:76-82``).
There is no model file or speech audio in the reference repo (``.gitignore:15-20``), so every
test, fixture and bench line uses these generators.  Pure numpy; no device code.
"""
import numpy as np

SAMPLE_RATE = 16000
CHUNK_SAMPLES = 1024          # 2048-byte chunks, runner.py:48


def stream_pcm(s: int, n_samples: int, kind: str = 'tone_noise') -> np.ndarray:
    """int16 PCM of stream ``s``.

    kinds:
      tone_noise  N(0, 3000^2) + 8000 sin(2 pi f_s t), f_s = 200 + 37 (s mod 97) Hz
      zeros       all-zero stream (exercises the eps clip in safe_log)
      square      full-scale +-32767/-32768 square wave, period 2*(20 + s mod 50) samples
      quiet       N(0, 2^2): a few LSBs of noise
    """
    rng = np.random.default_rng(1234 + s)
    t = np.arange(n_samples, dtype=np.float64) / SAMPLE_RATE
    if kind == 'tone_noise':
        f = 200.0 + 37.0 * (s % 97)
        x = rng.normal(0.0, 3000.0, n_samples) + 8000.0 * np.sin(2.0 * np.pi * f * t)
    elif kind == 'zeros':
        x = np.zeros(n_samples)
    elif kind == 'square':
        half = 20 + (s % 50)
        x = np.where((np.arange(n_samples) // half) % 2 == 0, 32767.0, -32768.0)
    elif kind == 'quiet':
        x = rng.normal(0.0, 2.0, n_samples)
    else:
        raise ValueError('unknown pcm kind: ' + kind)
    return np.clip(np.rint(x), -32768, 32767).astype('<i2')


def batch_pcm(n_streams: int, n_updates: int, chunk_samples: int = CHUNK_SAMPLES,
              kind: str = 'tone_noise', first_stream: int = 0) -> np.ndarray:
    """-> int16 [n_updates, n_streams, chunk_samples] (update-major: one C-ABI call consumes
    one contiguous [n_streams, chunk_samples] slab)."""
    out = np.empty((n_updates, n_streams, chunk_samples), dtype='<i2')
    for j in range(n_streams):
        out[:, j, :] = stream_pcm(first_stream + j, n_updates * chunk_samples, kind).reshape(
            n_updates, chunk_samples)
    return out


def _orthogonal(rng, rows, cols):
    a = rng.normal(0.0, 1.0, (max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    return q if rows >= cols else q.T


def make_weights(n_in: int = 13, units=(20,), seed: int = 42, dense_scale: float = 1.0) -> dict:
    """Random-init network in Keras layout (gate order z|r|h in the 3H columns):
       {'gru': [(kernel[F,3H], recurrent_kernel[H,3H], bias[3H]), ...],
        'dense_kernel': [H,1], 'dense_bias': [1]}    all float32."""
    rng = np.random.default_rng(seed)
    layers = []
    f = n_in
    for h in units:
        lim = np.sqrt(6.0 / (f + 3 * h))
        kernel = rng.uniform(-lim, lim, (f, 3 * h))
        rec = np.concatenate([_orthogonal(rng, h, h) for _ in range(3)], axis=1)
        bias = rng.normal(0.0, 0.1, 3 * h)
        layers.append((kernel.astype(np.float32), rec.astype(np.float32), bias.astype(np.float32)))
        f = h
    lim = np.sqrt(6.0 / (f + 1))
    dk = (dense_scale * rng.uniform(-lim, lim, (f, 1))).astype(np.float32)
    db = rng.normal(0.0, 0.1, 1).astype(np.float32)
    return {'gru': layers, 'dense_kernel': dk, 'dense_bias': db}
