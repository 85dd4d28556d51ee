"""
Raw network output -> calibrated confidence: drop-in for
``precise.threshold_decoder.ThresholdDecoder`` (/root/reference/precise/threshold_decoder.py:38-70).

Host-side scalar float64 work on ONE number per update; it is a step function of logit(raw)
(LUT index rounding), so parity tests compare the raw output and check decode() separately.
``decode_many`` is the vectorised form for batches of streams.
"""
import numpy as np

from .functions import asigmoid, sigmoid, pdf


class ThresholdDecoder:
    def __init__(self, mu_stds, center=0.5, resolution=200, min_z=-4, max_z=4):
        lows = [mu + min_z * std for mu, std in mu_stds]
        highs = [mu + max_z * std for mu, std in mu_stds]
        self.min_out = int(min(lows))
        self.max_out = int(max(highs))
        self.out_range = self.max_out - self.min_out
        self.cd = np.cumsum(self._calc_pd(mu_stds, resolution))
        self.center = center

    def _calc_pd(self, mu_stds, resolution):
        points = np.linspace(self.min_out, self.max_out, resolution * self.out_range)
        dens = np.sum([pdf(points, mu, std) for mu, std in mu_stds], axis=0)
        return dens / (resolution * len(mu_stds))

    def _scale(self, cp):
        if cp < self.center:
            return 0.5 * cp / self.center
        return 0.5 + 0.5 * (cp - self.center) / (1 - self.center)

    def decode(self, raw_output: float) -> float:
        if raw_output == 1.0 or raw_output == 0.0:      # saturated sigmoid passes through
            return raw_output
        if self.out_range == 0:
            cp = int(raw_output > self.min_out)
        else:
            ratio = (asigmoid(raw_output) - self.min_out) / self.out_range
            ratio = min(max(ratio, 0.0), 1.0)
            cp = self.cd[int(ratio * (len(self.cd) - 1) + 0.5)]
        return self._scale(cp)

    def decode_many(self, raw) -> np.ndarray:
        return np.array([self.decode(float(v)) for v in np.asarray(raw).reshape(-1)], dtype=np.float64)

    def encode(self, threshold: float) -> float:
        threshold = 0.5 * threshold / self.center
        if threshold < 0.5:
            cp = threshold * self.center * 2
        else:
            cp = (threshold - 0.5) * 2 * (1 - self.center) + self.center
        ratio = np.searchsorted(self.cd, cp) / len(self.cd)
        return sigmoid(self.min_out + self.out_range * ratio)
