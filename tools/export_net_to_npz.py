#!/usr/bin/env python3
"""
Export the weights of a Keras ``<model>.net`` file (HDF5; what precise-train writes through Keras'
ModelCheckpoint, /root/reference/precise/scripts/train.py:91-92, and KerasRunner loads,
/root/reference/precise/network_runner.py:77-95) to the ``.npz`` container the MI355X engine reads.

Run it wherever h5py is installed (the training machine); neither Keras nor TensorFlow is needed:

    python tools/export_net_to_npz.py hey-mycroft.net            # writes hey-mycroft.net.npz

``mycroft_precise_amd.model.load_weights('hey-mycroft.net')`` then picks the side-car up by itself, so
``Listener('hey-mycroft.net')`` keeps working with the reference's own file name (the ``.params`` file stays
``hey-mycroft.net.params``).  Since round 3 the package reads ``.net`` files itself (``mycroft_precise_amd/h5_model.py``,
no h5py); this exporter is the fallback for files that use HDF5 features outside that reader's subset.

Keras 2.x layout of a saved Sequential model: group ``model_weights`` (or the file root for weights-only
files) -> attribute ``layer_names`` -> one group per layer -> attribute ``weight_names`` -> datasets
``<layer>/<layer>/kernel:0`` etc.  The GRU of model.py:77-81 holds kernel [F,3H], recurrent_kernel [H,3H],
bias [3H] (gate order z|r|h; a [2,3H] bias would be a reset_after GRU, which the reference never builds).
"""
import sys

import numpy as np


def export(path, out=None):
    import h5py
    out = out or path + '.npz'
    with h5py.File(path, 'r') as f:
        root = f['model_weights'] if 'model_weights' in f else f
        layers = [n.decode() if isinstance(n, bytes) else n for n in root.attrs['layer_names']]
        gru, dense = [], None
        for name in layers:
            g = root[name]
            names = [n.decode() if isinstance(n, bytes) else n for n in g.attrs['weight_names']]
            arrays = {n.split('/')[-1].split(':')[0]: np.asarray(g[n], dtype=np.float32) for n in names}
            if 'recurrent_kernel' in arrays:
                if arrays['bias'].ndim != 1:
                    raise SystemExit('%s: layer %s is a reset_after GRU (bias %r): not a precise model' % (path, name, arrays['bias'].shape))
                gru.append((arrays['kernel'], arrays['recurrent_kernel'], arrays['bias']))
            elif 'kernel' in arrays:
                dense = (arrays['kernel'], arrays.get('bias', np.zeros(arrays['kernel'].shape[1], np.float32)))
        if not gru or dense is None:
            raise SystemExit('%s: expected GRU layer(s) followed by a Dense(1) layer, found %r' % (path, layers))
    # what the side-car was made from: load_weights refuses it once the .net file has been rewritten (retraining
    # through ModelCheckpoint / train_incremental rewrites <model>.net in place)
    import hashlib
    with open(path, 'rb') as fsrc:
        digest = hashlib.sha256(fsrc.read()).hexdigest()
    arrays = {'n_layers': np.int32(len(gru)), 'dense_kernel': dense[0], 'dense_bias': dense[1],
              'source_sha256': np.array(digest)}
    for i, (k, rk, b) in enumerate(gru):
        arrays['kernel_%d' % i], arrays['recurrent_kernel_%d' % i], arrays['bias_%d' % i] = k, rk, b
    with open(out, 'wb') as fo:
        np.savez(fo, **arrays)
    return out


if __name__ == '__main__':
    if len(sys.argv) not in (2, 3):
        sys.exit(__doc__)
    print('wrote', export(*sys.argv[1:]))
