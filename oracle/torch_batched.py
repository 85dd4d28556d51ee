"""
The oracle's batched restatement (oracle/listener.py: BatchedOracle) on torch CPU tensors.  TEST INFRASTRUCTURE ONLY.

Same arithmetic, same citations -- buffer_to_audio (/root/reference/precise/util.py:35-37), Listener.update_vectors
(/root/reference/precise/network_runner.py:125-146), sonopy.mfcc_spec restated (oracle/sonopy_restated.py: Q1-Q7), the Keras
GRU forward restated (oracle/keras_gru.py: K1-K8) -- vectorised over B lock-step streams, with torch's multi-threaded CPU
kernels (pocketfft / MKL-style GEMMs) underneath instead of one numpy process per core.  It exists for ONE purpose: BASELINE.md
section 2 promised the batched CPU baseline B2 as "numpy / torch-CPU", and bench.py's ``cpu_baseline`` keeps whichever of the
two is faster on the GPU box's host (labelled).  tests/test_oracle.py pins it to the numpy oracle (<= 1e-6 on the probability,
<= 1e-9 on the features): parity stays anchored on the numpy restatement and the golden fixtures.
"""
import numpy as np
import torch

from . import sonopy_restated as sonopy
from .listener import Params


class TorchBatchedOracle:
    def __init__(self, weights, n_streams: int, pr: Params = None, threads: int = None):
        if threads:
            torch.set_num_threads(int(threads))
        self.pr = pr or Params()
        if self.pr.vectorizer == 3 or self.pr.use_delta:
            raise NotImplementedError('the torch baseline covers the stock configuration (sonopy mfccs, no deltas)')
        self.n = int(n_streams)
        pr = self.pr
        bins = pr.n_fft // 2 + 1
        self.filt_t = torch.from_numpy(np.ascontiguousarray(sonopy.filterbanks(pr.sample_rate, pr.n_filt, bins).T))      # [bins, n_filt] float64
        n = np.arange(pr.n_filt)
        dct = np.stack([np.cos(np.pi * c * (2 * n + 1) / (2.0 * pr.n_filt)) * (np.sqrt(1.0 / pr.n_filt) if c == 0 else np.sqrt(2.0 / pr.n_filt))
                        for c in range(pr.n_mfcc)], axis=1)                  # scipy.fftpack.dct(type=2, norm='ortho') as a matrix [n_filt, n_mfcc]
        self.dct = torch.from_numpy(np.ascontiguousarray(dct))
        self.layers = [(torch.from_numpy(np.ascontiguousarray(k, dtype=np.float32)), torch.from_numpy(np.ascontiguousarray(rk, dtype=np.float32)),
                        torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32))) for k, rk, b in weights['gru']]
        self.dk = torch.from_numpy(np.ascontiguousarray(weights['dense_kernel'], dtype=np.float32).reshape(-1, 1))
        self.db = float(np.asarray(weights['dense_bias'], dtype=np.float32).reshape(-1)[0])
        self.eps = float(np.finfo(float).eps)
        self.clear()

    def clear(self):
        self.window_audio = torch.zeros((self.n, 0), dtype=torch.float64)
        self.mfccs = torch.zeros((self.n, self.pr.n_features, self.pr.n_mfcc), dtype=torch.float64)

    def update_vectors(self, pcm) -> torch.Tensor:
        pr = self.pr
        audio = (torch.as_tensor(pcm).to(torch.float32) / 32768.0).to(torch.float64)              # util.py:35-37, then float64 like window_audio
        self.window_audio = torch.cat((self.window_audio, audio), dim=1)
        length = self.window_audio.shape[1]
        if length >= pr.window_samples:
            n_new = 1 + (length - pr.window_samples) // pr.hop_samples
            frames = self.window_audio.unfold(1, pr.window_samples, pr.hop_samples)[:, :n_new, :pr.n_fft]        # Q2: rfft(n=512) crops the window
            spec = torch.fft.rfft(frames, n=pr.n_fft)
            powers = (spec.real ** 2 + spec.imag ** 2) / pr.n_fft
            mels = torch.log(torch.clamp(powers @ self.filt_t, min=self.eps))
            new = mels @ self.dct
            new[..., 0] = torch.log(torch.clamp(powers.sum(-1), min=self.eps))                    # Q7
            self.window_audio = self.window_audio[:, n_new * pr.hop_samples:].contiguous()
            if n_new > pr.n_features:
                new = new[:, -pr.n_features:]
            self.mfccs = torch.cat((self.mfccs[:, new.shape[1]:], new), dim=1)
        return self.mfccs

    def predict(self, feats: torch.Tensor) -> torch.Tensor:
        a = feats.to(torch.float32)
        for li, (k, rk, b) in enumerate(self.layers):
            H = rk.shape[0]
            xw = a @ k + b                                                                         # [B, T, 3H]: the input projections of every timestep at once
            h = torch.zeros((a.shape[0], H), dtype=torch.float32)
            outs = []
            for t in range(a.shape[1]):
                zr = xw[:, t, :2 * H] + h @ rk[:, :2 * H]
                zr = torch.clamp(0.2 * zr + 0.5, 0.0, 1.0)                                        # K: TF hard_sigmoid
                z, r = zr[:, :H], zr[:, H:]
                hh = xw[:, t, 2 * H:] + (r * h) @ rk[:, 2 * H:]                                   # activation='linear' (model.py:77-81)
                h = z * h + (1.0 - z) * hh
                if li + 1 < len(self.layers):
                    outs.append(h)
            a = torch.stack(outs, dim=1) if li + 1 < len(self.layers) else h
        return torch.sigmoid(a @ self.dk + self.db)[:, 0]

    def update_raw(self, pcm) -> np.ndarray:
        return self.predict(self.update_vectors(pcm)).numpy()
