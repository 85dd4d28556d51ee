"""
Raw network output -> calibrated confidence: API-compatible stand-in for
``precise.threshold_decoder.ThresholdDecoder`` (/root/reference/precise/threshold_decoder.py:38-70).

The network's sigmoid output is modelled as logit-normal: ``mu_stds`` lists (mean, std) pairs of
logit(output) measured on positive samples (``precise-calc-threshold``).  Their summed normal
densities, sampled ``resolution`` times per logit unit between the smallest ``mu + min_z*std`` and the
largest ``mu + max_z*std`` (both truncated to int), are accumulated into the lookup table ``cd``; a
raw output is mapped to its logit, to a table index (rounded to nearest), to a cumulative probability,
and finally stretched so that ``center`` lands on 0.5.

Host-side scalar work on ONE number per prediction (``decode``); ``pe_decode`` runs the same arithmetic for
every stream on the device.  The logit is evaluated in the scalar type of the argument, exactly as the
reference's ``asigmoid`` (functions.py:99-101) does: the runners return a numpy float32 scalar, for which
``1 / x - 1`` is float32 arithmetic, and only the logarithm is taken in float64.  ``decode(np.float32(p))``
and the device decoder therefore return what ``Listener.update`` returns in the reference; ``decode(float(p))``
is the float64 evaluation and can land in the neighbouring table bin.
"""
import numpy as np

from .functions import asigmoid, sigmoid, pdf


class ThresholdDecoder:
    def __init__(self, mu_stds, center=0.5, resolution=200, min_z=-4, max_z=4):
        self.min_out = int(min(mu + min_z * std for mu, std in mu_stds))
        self.max_out = int(max(mu + max_z * std for mu, std in mu_stds))
        self.out_range = self.max_out - self.min_out
        self.center = center
        self.cd = np.cumsum(self._calc_pd(mu_stds, resolution))

    def _calc_pd(self, mu_stds, resolution):
        """Average of the normal densities on the logit grid, scaled to integrate to ~1."""
        grid = np.linspace(self.min_out, self.max_out, resolution * self.out_range)
        total = np.sum([pdf(grid, mu, std) for mu, std in mu_stds], axis=0)
        return total / (resolution * len(mu_stds))

    def _stretch(self, cp):
        """Piecewise-linear map [0, center] -> [0, 0.5], [center, 1] -> [0.5, 1]."""
        c = self.center
        return 0.5 * cp / c if cp < c else 0.5 + 0.5 * (cp - c) / (1 - c)

    def _cumulative(self, raw_output):
        if self.out_range == 0:                       # degenerate calibration: a plain step
            return int(raw_output > self.min_out)
        position = (asigmoid(raw_output) - self.min_out) / self.out_range
        position = min(max(position, 0.0), 1.0)
        return self.cd[int(position * (len(self.cd) - 1) + 0.5)]

    def decode(self, raw_output: float) -> float:
        """Confidence of one raw network output; exactly saturated outputs pass through."""
        if raw_output == 1.0 or raw_output == 0.0:
            return raw_output
        return self._stretch(self._cumulative(raw_output))

    def decode_many(self, raw) -> np.ndarray:
        """Element-wise ``decode``; every element keeps the array's scalar type (a float32 array decodes as
        the reference decodes a float32 network output, see ``decode``)."""
        return np.array([self.decode(v) for v in np.asarray(raw).reshape(-1)], dtype=np.float64)

    def encode(self, threshold: float) -> float:
        """Inverse direction: the raw network output whose decoded confidence is ``threshold``."""
        c = self.center
        t = 0.5 * threshold / c
        cp = t * c * 2 if t < 0.5 else (t - 0.5) * 2 * (1 - c) + c
        position = np.searchsorted(self.cd, cp) / len(self.cd)
        return sigmoid(self.min_out + self.out_range * position)
