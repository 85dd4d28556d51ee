"""
Network description and weight files: the inference-side counterpart of ``precise.model``
(/root/reference/precise/model.py:28-91).

The reference builds ``Sequential([GRU(units, activation='linear', name='net'), Dense(1,
'sigmoid')])`` (model.py:76-82) and stores it as Keras ``.net`` (HDF5) or frozen-graph ``.pb``.
Neither TensorFlow, Keras nor h5py exist on the target, and the reference ships no model file,
so this module defines the weights as plain arrays in Keras layout:

    {'gru': [(kernel[F,3H], recurrent_kernel[H,3H], bias[3H]), ...],   gate order z | r | h
     'dense_kernel': [H,1], 'dense_bias': [1]}

stored as ``<model>.npz`` next to the usual ``<model>.npz.params`` JSON, read straight from the
reference's frozen-graph ``<model>.pb`` (``pb_model.py``: a protobuf wire reader, no TensorFlow), or from a
Keras ``<model>.net`` (``h5_model.py``: an HDF5 reader for what h5py / Keras write, no h5py; files outside that
subset fall back to the side-car ``<model>.net.npz`` that ``tools/export_net_to_npz.py`` writes).
"""
import numpy as np

from .params import inject_params, pr


class ModelParams:
    """Same fields and defaults as the reference's attrs class (model.py:28-45); only
    ``recurrent_units`` matters for inference."""

    def __init__(self, recurrent_units=20, dropout=0.2, extra_metrics=False, skip_acc=False,
                 loss_bias=0.7, freeze_till=0):
        self.recurrent_units = recurrent_units
        self.dropout = dropout
        self.extra_metrics = extra_metrics
        self.skip_acc = skip_acc
        self.loss_bias = loss_bias
        self.freeze_till = freeze_till


def save_weights(model_name: str, weights: dict):
    arrays = {'n_layers': np.int32(len(weights['gru'])),
              'dense_kernel': np.asarray(weights['dense_kernel'], np.float32),
              'dense_bias': np.asarray(weights['dense_bias'], np.float32)}
    for i, (k, rk, b) in enumerate(weights['gru']):
        arrays['kernel_%d' % i] = np.asarray(k, np.float32)
        arrays['recurrent_kernel_%d' % i] = np.asarray(rk, np.float32)
        arrays['bias_%d' % i] = np.asarray(b, np.float32)
    with open(model_name, 'wb') as f:        # np.savez would append '.npz' to other extensions
        np.savez(f, **arrays)


def _check_sidecar(net_file: str, sidecar: str):
    """The side-car must have been exported from THIS .net file: retraining (ModelCheckpoint, train_incremental)
    rewrites ``<model>.net`` in place, and serving the old export would be silent.  Exports carry the sha256 of
    their source; older ones are compared by modification time."""
    import hashlib
    import warnings
    from os.path import getmtime, isfile
    if not isfile(net_file):
        return                                       # only the export travelled to this machine
    with np.load(sidecar, allow_pickle=False) as z:
        want = str(z['source_sha256']) if 'source_sha256' in z.files else None
    if want is not None:
        with open(net_file, 'rb') as f:
            have = hashlib.sha256(f.read()).hexdigest()
        if have != want:
            raise ValueError('%s was exported from another version of %s (the model was rewritten since): run '
                             '`python tools/export_net_to_npz.py %s` again' % (sidecar, net_file, net_file))
    elif getmtime(net_file) > getmtime(sidecar):
        warnings.warn('%s is newer than its exported weights %s: re-run tools/export_net_to_npz.py' % (net_file, sidecar))


def _sidecar_matches(net_file: str, sidecar: str) -> bool:
    import hashlib
    try:
        with np.load(sidecar, allow_pickle=False) as z:
            if 'source_sha256' not in z.files:
                return False
            want = str(z['source_sha256'])
        with open(net_file, 'rb') as f:
            return hashlib.sha256(f.read()).hexdigest() == want
    except Exception:        # a truncated / corrupt side-car (zipfile.BadZipFile, EOFError, KeyError, ...) is simply "no match":
        return False         # the native .net reader stays the fallback


def load_weights(model_name: str) -> dict:
    if model_name.endswith('.pb'):
        from .pb_model import weights_from_pb
        return weights_from_pb(model_name)
    if model_name.endswith('.net'):
        # Keras HDF5: read directly (h5_model.py: the subset of the format that h5py / Keras write); a file that uses
        # something else, or is not there at all, falls back to the side-car that tools/export_net_to_npz.py writes
        # next to it where h5py exists
        from os.path import isfile
        from .h5_model import H5FormatError, H5Unsupported, weights_from_net
        why = 'the file does not exist'
        sidecar = model_name + '.npz'
        if isfile(model_name) and isfile(sidecar) and _sidecar_matches(model_name, sidecar):
            # a side-car that h5py exported from exactly this file (sha256 recorded inside) outranks the spec-written
            # reader, which no libhdf5-written file has pinned yet (h5_model.py header)
            pass
        else:
            if isfile(model_name):
                try:
                    return weights_from_net(model_name)
                except (H5Unsupported, H5FormatError) as ex:
                    why = str(ex)
            if not isfile(sidecar):
                raise NotImplementedError(
                    'cannot read %s (%s) and its exported weights %s do not exist: run `python tools/export_net_to_npz.py '
                    '%s` on a machine with h5py, or freeze the model with precise-convert to .pb' % (model_name, why, sidecar, model_name))
            _check_sidecar(model_name, sidecar)
        model_name = sidecar
    with np.load(model_name, allow_pickle=False) as z:
        n = int(z['n_layers'])
        layers = [(z['kernel_%d' % i], z['recurrent_kernel_%d' % i], z['bias_%d' % i]) for i in range(n)]
        return {'gru': layers, 'dense_kernel': z['dense_kernel'], 'dense_bias': z['dense_bias']}


def load_precise_model(model_name: str) -> dict:
    """Reference name (model.py:48-54): inject the model's params, return its weights."""
    inject_params(model_name)
    return load_weights(model_name)


def create_model(model_name, params: ModelParams = None, seed: int = 42) -> dict:
    """Load ``model_name`` if it exists, else a random-init network of the reference's topology
    for the current ``pr`` (model.py:57-91, forward pass only)."""
    from os.path import isfile
    from .synth import make_weights
    if model_name and isfile(model_name):
        return load_precise_model(model_name)
    params = params or ModelParams()
    return make_weights(n_in=pr.feature_size, units=(params.recurrent_units,), seed=seed)
