import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def stock_weights():
    g = golden('weights_stock_seed42.npz')
    return {'gru': [(g['kernel'], g['recurrent_kernel'], g['bias'])],
            'dense_kernel': g['dense_kernel'], 'dense_bias': g['dense_bias']}
