"""
Audio-pipeline parameters: drop-in for ``precise.params``
(/root/reference/precise/params.py:28-165): ``ListenerParams``, ``Vectorizer``, the process-global
``pr``, ``inject_params`` / ``save_params`` and the ``<model>.params`` JSON overlay.

Same field names, defaults, derived sizes and "last inject wins" global semantics as the
reference, but written as a plain mutable-dict class; engines snapshot the sizes they need at
construction (``pe_params``), so later injections never disturb a running engine.
"""
import hashlib
import json
from math import floor
from os.path import isfile

_FIELDS = ('buffer_t', 'window_t', 'hop_t', 'sample_rate', 'sample_depth', 'n_fft', 'n_filt',
           'n_mfcc', 'use_delta', 'vectorizer', 'threshold_config', 'threshold_center')


class Vectorizer:
    """Which function vectorizes audio (params.py:121-132)."""
    mels = 1
    mfccs = 2
    speechpy_mfccs = 3


class ListenerParams:
    def __init__(self, buffer_t, window_t, hop_t, sample_rate, sample_depth, n_fft, n_filt, n_mfcc,
                 use_delta, vectorizer, threshold_config, threshold_center):
        values = locals()
        for name in _FIELDS:
            self.__dict__[name] = values[name]

    def __setattr__(self, name, value):        # frozen like the attrs class; update via __dict__
        raise AttributeError('ListenerParams is frozen; use inject_params()')

    def __repr__(self):
        return 'ListenerParams(%s)' % ', '.join('%s=%r' % (k, self.__dict__[k]) for k in _FIELDS)

    def copy(self):
        return ListenerParams(**{k: self.__dict__[k] for k in _FIELDS})

    # derived sizes (params.py:73-109)
    @property
    def hop_samples(self):
        return int(self.sample_rate * self.hop_t + 0.5)

    @property
    def window_samples(self):
        return int(self.sample_rate * self.window_t + 0.5)

    @property
    def buffer_samples(self):
        total = int(self.sample_rate * self.buffer_t + 0.5)
        return self.hop_samples * (total // self.hop_samples)

    @property
    def n_features(self):
        return 1 + int(floor((self.buffer_samples - self.window_samples) / self.hop_samples))

    @property
    def max_samples(self):
        return int(self.buffer_t * self.sample_rate)

    @property
    def feature_size(self):
        base = self.n_filt if self.vectorizer == Vectorizer.mels else self.n_mfcc
        return 2 * base if self.use_delta else base

    def vectorization_md5_hash(self):
        keys = sorted(k for k in pr.__dict__ if k not in ('threshold_config', 'threshold_center'))
        return hashlib.md5(str([pr.__dict__[k] for k in keys]).encode()).hexdigest()


# process-global defaults (params.py:140-144)
pr = ListenerParams(buffer_t=1.5, window_t=0.1, hop_t=0.05, sample_rate=16000, sample_depth=2,
                    n_fft=512, n_filt=20, n_mfcc=13, use_delta=False,
                    threshold_config=((6, 4),), threshold_center=0.2, vectorizer=Vectorizer.mfccs)

# old .params files without the newer keys get these (params.py:147)
compatibility_params = dict(vectorizer=Vectorizer.speechpy_mfccs)


def inject_params(model_name: str) -> ListenerParams:
    """Overlay ``<model_name>.params`` (JSON) onto the global ``pr`` (params.py:150-159)."""
    path = model_name + '.params'
    try:
        with open(path) as f:
            loaded = json.load(f)
        pr.__dict__.update(compatibility_params, **loaded)
    except (OSError, ValueError, TypeError):
        if isfile(model_name):
            print('Warning: Failed to load parameters from ' + path)
    return pr


def save_params(model_name: str):
    """Write the global ``pr`` next to the model (params.py:162-165)."""
    with open(model_name + '.params', 'w') as f:
        json.dump(pr.__dict__, f)
