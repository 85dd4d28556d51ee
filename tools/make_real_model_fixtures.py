#!/usr/bin/env python3
"""
Pin the model-file readers against the REAL libraries (maintainer tool; needs h5py, optionally Keras / TensorFlow).

`mycroft_precise_amd/h5_model.py` (Keras `.net` = HDF5) and `mycroft_precise_amd/pb_model.py` (frozen GraphDef `.pb`)
were written from the published format specifications in a container without libhdf5 / h5py / TensorFlow; the tests read
files from `tests/h5_writer.py` and from a google.protobuf-encoded GraphDef.  This script produces the missing evidence
wherever the real packages exist, and drops it into `tests/golden/` with provenance:

    python tools/make_real_model_fixtures.py            # writes whatever the installed packages allow

    tests/golden/real_h5py_model.net        h5py-written file in the Keras 2.x layout (model.py:76-82 topology)
    tests/golden/real_keras_model.net       keras.models.Sequential(...).save()   (train.py:91-92)      [Keras present]
    tests/golden/real_tf_model.pb           convert_variables_to_constants / graph_util output (convert.py:59-81) [TF present]
    tests/golden/real_model_weights.npz     the weights every one of them must yield + a `provenance` string array

`tests/test_host.py::test_readers_against_committed_real_fixtures_when_present` then reads them back on every
machine, with no h5py / TensorFlow needed; `test_readers_against_live_h5py_tensorflow_when_importable` does the
same in-process where the packages are importable.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def write_h5py_net(path, weights, **dataset_kw):
    """The Keras 2.2.4 layout (keras/engine/saving.py) written with h5py itself: what libhdf5 puts on disk."""
    import h5py
    import json
    k, rk, b = weights['gru'][0]
    with h5py.File(path, 'w') as f:
        f.attrs['keras_version'] = b'2.2.4'
        f.attrs['backend'] = b'tensorflow'
        f.attrs['model_config'] = json.dumps({'class_name': 'Sequential', 'config': {'name': 'sequential_1', 'layers': [
            {'class_name': 'GRU', 'config': {'name': 'net', 'units': int(rk.shape[0]), 'activation': 'linear'}},
            {'class_name': 'Dense', 'config': {'name': 'dense_1', 'units': 1, 'activation': 'sigmoid'}}]}}).encode('utf8')
        mw = f.create_group('model_weights')
        mw.attrs['layer_names'] = [b'net', b'dense_1']
        mw.attrs['backend'] = b'tensorflow'
        mw.attrs['keras_version'] = b'2.2.4'
        g = mw.create_group('net')
        g.attrs['weight_names'] = [b'net/kernel:0', b'net/recurrent_kernel:0', b'net/bias:0']
        for name, arr in (('net/kernel:0', k), ('net/recurrent_kernel:0', rk), ('net/bias:0', b)):
            g.create_dataset(name, data=np.asarray(arr, np.float32), **dataset_kw)
        d = mw.create_group('dense_1')
        d.attrs['weight_names'] = [b'dense_1/kernel:0', b'dense_1/bias:0']
        d.create_dataset('dense_1/kernel:0', data=np.asarray(weights['dense_kernel'], np.float32), **dataset_kw)
        d.create_dataset('dense_1/bias:0', data=np.asarray(weights['dense_bias'], np.float32).reshape(-1), **dataset_kw)
    return 'h5py %s / libhdf5 %s' % (h5py.__version__, h5py.version.hdf5_version)


def write_keras_net(path, weights):
    """The reference's own model (model.py:76-82) saved by Keras (train.py:91-92)."""
    try:
        from keras.models import Sequential
        from keras.layers import GRU, Dense
        import keras
    except ImportError:
        from tensorflow.keras.models import Sequential
        from tensorflow.keras.layers import GRU, Dense
        from tensorflow import keras
    k, rk, b = weights['gru'][0]
    model = Sequential()
    model.add(GRU(int(rk.shape[0]), activation='linear', input_shape=(29, int(k.shape[0])), dropout=0.2, name='net', reset_after=False))
    model.add(Dense(1, activation='sigmoid'))
    model.layers[0].set_weights([k, rk, b])
    model.layers[1].set_weights([np.asarray(weights['dense_kernel']), np.asarray(weights['dense_bias']).reshape(-1)])
    model.save(path, save_format='h5') if 'save_format' in model.save.__code__.co_varnames else model.save(path)
    return 'keras %s' % keras.__version__


def write_tf_pb(path, weights):
    """A frozen GraphDef whose Const nodes carry the weights under the names precise-convert leaves (convert.py:59-81)."""
    import tensorflow as tf
    tf1 = tf.compat.v1 if hasattr(tf, 'compat') and hasattr(tf.compat, 'v1') else tf
    g = tf1.Graph()
    k, rk, b = weights['gru'][0]
    with g.as_default():
        tf1.placeholder(tf.float32, [None, 29, int(k.shape[0])], name='net_input')
        for name, arr in (('net/kernel', k), ('net/recurrent_kernel', rk), ('net/bias', b),
                          ('dense_1/kernel', weights['dense_kernel']), ('dense_1/bias', np.asarray(weights['dense_bias']).reshape(-1))):
            tf.constant(np.asarray(arr, np.float32), name=name)
    with open(path, 'wb') as f:
        f.write(g.as_graph_def().SerializeToString())
    return 'tensorflow %s' % tf.__version__


def main():
    from mycroft_precise_amd import synth
    weights = synth.make_weights()
    prov = []
    made = []
    for label, fn, out in (('h5py', write_h5py_net, 'real_h5py_model.net'), ('keras', write_keras_net, 'real_keras_model.net'),
                           ('tensorflow', write_tf_pb, 'real_tf_model.pb')):
        try:
            prov.append('%s: %s' % (out, fn(os.path.join(GOLDEN, out), weights)))
            made.append(out)
        except ImportError as ex:
            print('skipped %s (%s)' % (out, ex))
    if not made:
        raise SystemExit('neither h5py nor Keras nor TensorFlow is importable here: nothing written')
    k, rk, b = weights['gru'][0]
    np.savez(os.path.join(GOLDEN, 'real_model_weights.npz'), kernel=k, recurrent_kernel=rk, bias=b,
             dense_kernel=weights['dense_kernel'], dense_bias=weights['dense_bias'], provenance=np.array(prov))
    print('wrote', ', '.join(made + ['real_model_weights.npz']))


if __name__ == '__main__':
    main()
