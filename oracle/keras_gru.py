"""
CPU restatement of the network the reference builds in
``/root/reference/precise/model.py:76-82``:

    Sequential([GRU(units, activation='linear', input_shape=(n_features, feature_size),
                    dropout=..., name='net'),
                Dense(1, activation='sigmoid')])

executed by ``TensorFlowRunner.predict`` / ``KerasRunner.predict``
(``/root/reference/precise/network_runner.py:69-74,88-95``).  TEST INFRASTRUCTURE ONLY.

The layer arithmetic lives in third-party Keras 2.2.4 / TensorFlow 1.13.1
(``requirements.txt:11,38``), absent from /root/reference and not installable here, so
this restates Keras' published ``GRUCell.call`` for the constructor defaults the reference
relies on.  **Parity with real Keras/TF bits is UNPINNED.**  Named assumptions:

  K1  ``recurrent_activation='hard_sigmoid'``: hs(v) = clip(0.2*v + 0.5, 0, 1)
  K2  ``reset_after=False`` (r is applied to h BEFORE the recurrent matmul),
      ``use_bias=True`` with ONE bias vector of length 3H
  K3  gate order inside kernel / recurrent_kernel / bias columns: z | r | h
  K4  ``activation='linear'`` [REF model.py:78]: candidate state is NOT squashed
  K5  h0 = 0 on every ``predict`` call (stateless layer); dropout inactive at inference
  K6  recurrence:  z  = hs(x W_z + b_z + h U_z)
                   r  = hs(x W_r + b_r + h U_r)
                   hh =    x W_h + b_h + (r*h) U_h
                   h' = z*h + (1-z)*hh
  K7  head: p = 1 / (1 + exp(-(h_T . W_d + b_d)))
  K8  everything float32 (the feed casts the float64 features to float32).

Stacked GRUs (BASELINE config 4, not in the reference) chain layers with
``return_sequences=True`` on all but the last.
"""
import numpy as np


def hard_sigmoid(v):
    return np.clip(np.float32(0.2) * v + np.float32(0.5), np.float32(0.0), np.float32(1.0))


def gru_layer(x, kernel, recurrent_kernel, bias, return_sequences=False, dtype=np.float32):
    """x: [N, T, F] -> [N, H] (or [N, T, H]).  K1-K6."""
    x = np.asarray(x, dtype=dtype)
    W = np.asarray(kernel, dtype=dtype)
    U = np.asarray(recurrent_kernel, dtype=dtype)
    b = np.asarray(bias, dtype=dtype)
    n, t_len, _ = x.shape
    hid = U.shape[0]
    Wz, Wr, Wh = W[:, :hid], W[:, hid:2 * hid], W[:, 2 * hid:]
    Uz, Ur, Uh = U[:, :hid], U[:, hid:2 * hid], U[:, 2 * hid:]
    bz, br, bh = b[:hid], b[hid:2 * hid], b[2 * hid:]
    h = np.zeros((n, hid), dtype=dtype)
    seq = np.empty((n, t_len, hid), dtype=dtype) if return_sequences else None
    if dtype == np.float32:
        hs = hard_sigmoid
    else:
        hs = lambda v: np.clip(0.2 * v + 0.5, 0.0, 1.0)
    one = dtype(1.0)
    for t in range(t_len):
        xt = x[:, t, :]
        z = hs(xt @ Wz + bz + h @ Uz)
        r = hs(xt @ Wr + br + h @ Ur)
        hh = xt @ Wh + bh + (r * h) @ Uh
        h = z * h + (one - z) * hh
        if return_sequences:
            seq[:, t, :] = h
    return seq if return_sequences else h


def predict(x, weights, dtype=np.float32):
    """
    x: [N, T, F];  weights: dict with
        'gru': list of (kernel[F_l, 3H_l], recurrent_kernel[H_l, 3H_l], bias[3H_l])
        'dense_kernel': [H_last, 1], 'dense_bias': [1]
    -> [N, 1] float32 raw sigmoid outputs   (Runner.predict contract,
       network_runner.py:35-38)
    """
    a = np.asarray(x, dtype=dtype)
    layers = weights['gru']
    for li, (k, rk, b) in enumerate(layers):
        a = gru_layer(a, k, rk, b, return_sequences=(li + 1 < len(layers)), dtype=dtype)
    logit = a @ np.asarray(weights['dense_kernel'], dtype=dtype) + np.asarray(weights['dense_bias'], dtype=dtype)
    return (dtype(1.0) / (dtype(1.0) + np.exp(-logit))).astype(dtype)


class NumpyRunner:
    """Drop-in for the reference's ``Runner`` ABC (network_runner.py:31-42), injected via
    ``Listener(..., runner_cls=...)``.  ``weights`` is bound with ``make_runner_cls``."""
    weights = None

    def __init__(self, model_name):
        self.model_name = model_name

    def predict(self, inputs):
        return predict(inputs, self.weights)

    def run(self, inp):
        return self.predict(np.asarray(inp)[np.newaxis])[0][0]


def make_runner_cls(weights):
    return type('BoundNumpyRunner', (NumpyRunner,), {'weights': weights})
