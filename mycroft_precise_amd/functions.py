"""Scalar helpers of /root/reference/precise/functions.py:94-108 used by ThresholdDecoder."""
import math

import numpy as np


def sigmoid(x):
    return 1 / (1 + math.exp(-x))


def asigmoid(x):
    """logit; ZeroDivisionError at 0 and ValueError at 1, as the reference's expression."""
    return -math.log(1 / x - 1)


def pdf(x, mu, std):
    if std == 0:
        return 0
    return (1.0 / (std * math.sqrt(2 * math.pi))) * np.exp(-(x - mu) ** 2 / (2 * std ** 2))
