"""
Weights out of (and into) the frozen-graph ``.pb`` files the reference runs
(/root/reference/precise/scripts/convert.py:59-81 writes them,
/root/reference/precise/network_runner.py:53-67 loads them) -- WITHOUT TensorFlow.

``precise-convert`` freezes the Keras model with ``convert_variables_to_constants``: every layer
variable becomes a ``Const`` node named ``<layer>/<variable>`` holding a float32 ``TensorProto``.
For the network of ``model.py:76-82`` that is ``net/kernel`` [F,3H], ``net/recurrent_kernel``
[H,3H], ``net/bias`` [3H] (the GRU layer is named 'net', model.py:80) and ``dense_N/kernel``
[H,1], ``dense_N/bias`` [1].  This module is a minimal protobuf wire-format reader that walks
``GraphDef.node[*]`` and decodes exactly those tensors; everything else in the graph (the TF
``while`` loop that implements the GRU) is ignored -- the HIP kernels replace it.

Field numbers used (tensorflow/core/framework/*.proto):
    GraphDef.node = 1            NodeDef.name = 1, .op = 2, .attr = 5 (map<string, AttrValue>)
    map entry: key = 1, value = 2        AttrValue.tensor = 8
    TensorProto.dtype = 1 (DT_FLOAT = 1), .tensor_shape = 2, .tensor_content = 4, .float_val = 5
    TensorShapeProto.dim = 2             Dim.size = 1

The writer produces a small but well-formed frozen GraphDef (Placeholder ``net_input``, the Const
nodes, Identity ``net_output``) so that weights can be exchanged in the reference's container; it
is what the tests read back, and real ``.pb`` files produced by TensorFlow use the same encoding.
No real ``.pb`` ships with the reference (``.gitignore:15-20``), so reading one produced by
TensorFlow itself is untested here.
"""
import struct

import numpy as np

DT_FLOAT = 1
MAX_TENSOR_ELEMENTS = 1 << 26          # 64 M floats: no layer of a wake-word model comes near (damaged dims must not allocate)


# ---- protobuf wire format ---------------------------------------------------------------------
def _varint(buf, pos):
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise ValueError('truncated protobuf message (varint runs past the end)')
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _fields(buf):
    """Yield (field_number, wire_type, value) for one message; value is int or a memoryview."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            val = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        if pos > end:
            raise ValueError('truncated protobuf message')
        yield num, wt, val


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_field(num, payload):
    """length-delimited field"""
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_int(num, v):
    return _enc_varint(num << 3) + _enc_varint(v)


# ---- decoding -----------------------------------------------------------------------------------
def _decode_tensor(buf):
    dtype, dims, content, floats = None, [], None, []
    for num, wt, val in _fields(buf):
        if num == 1 and wt == 0:
            dtype = val
        elif num == 2 and wt == 2:
            for n2, w2, v2 in _fields(val):
                if n2 == 2 and w2 == 2:
                    size = 0
                    for n3, w3, v3 in _fields(v2):
                        if n3 == 1 and w3 == 0:
                            size = v3
                    dims.append(size)
        elif num == 4 and wt == 2:
            content = bytes(val)
        elif num == 5:
            if wt == 2:                                     # packed repeated float
                floats.extend(struct.unpack('<%df' % (len(val) // 4), bytes(val)))
            elif wt == 5:
                floats.append(struct.unpack('<f', val)[0])
    if dtype != DT_FLOAT:
        return None
    count = 1
    for dim in dims:
        count *= int(dim)
        if dim < 0 or count > MAX_TENSOR_ELEMENTS:
            raise ValueError('float tensor of shape %r' % (dims,))
    if content is not None and len(content) == 4 * count:
        arr = np.frombuffer(content, dtype='<f4').copy()
    elif len(floats) == count:
        arr = np.array(floats, dtype=np.float32)
    elif len(floats) == 1:                                   # TF stores constant-filled tensors as one value
        arr = np.full(count, floats[0], dtype=np.float32)
    else:
        raise ValueError('float tensor with %d values for shape %r' % (len(floats), dims))
    return arr.reshape(dims)


def read_const_tensors(path):
    """{node name: float32 ndarray} for every float ``Const`` node of a GraphDef file.  A damaged file raises
    ValueError, whatever the parser tripped over."""
    try:
        return _read_const_tensors(path)
    except ValueError:
        raise
    except (IndexError, TypeError, struct.error, UnicodeDecodeError, OverflowError, MemoryError) as ex:
        raise ValueError('%s: damaged or not a GraphDef (%s: %s)' % (path, type(ex).__name__, ex)) from ex


def _read_const_tensors(path):
    with open(path, 'rb') as f:
        data = memoryview(f.read())
    out = {}
    for num, wt, node in _fields(data):
        if num != 1 or wt != 2:
            continue
        name = op = None
        tensor = None
        for n2, w2, v2 in _fields(node):
            if n2 == 1 and w2 == 2:
                name = bytes(v2).decode()
            elif n2 == 2 and w2 == 2:
                op = bytes(v2).decode()
            elif n2 == 5 and w2 == 2:                        # attr map entry
                key = value = None
                for n3, w3, v3 in _fields(v2):
                    if n3 == 1 and w3 == 2:
                        key = bytes(v3).decode()
                    elif n3 == 2 and w3 == 2:
                        value = v3
                if key == 'value' and value is not None:
                    for n4, w4, v4 in _fields(value):
                        if n4 == 8 and w4 == 2:
                            tensor = v4
        if op == 'Const' and name and tensor is not None:
            arr = _decode_tensor(tensor)
            if arr is not None:
                out[name] = arr
    return out


def weights_from_pb(path):
    """Keras-layout weight dict (see model.py) from a frozen precise ``.pb``."""
    consts = read_const_tensors(path)
    strip = {k[len('import/'):] if k.startswith('import/') else k: v for k, v in consts.items()}
    rec = sorted(k for k in strip if k.endswith('/recurrent_kernel'))
    if not rec:
        raise ValueError('%s holds no */recurrent_kernel constant: not a frozen precise GRU model' % path)
    layers, used = [], set()
    # 'net' first, then any further GRU layers in name order (stacked models)
    rec.sort(key=lambda k: (k.split('/')[0] != 'net', k))
    for rk_name in rec:
        base = rk_name[:-len('/recurrent_kernel')]
        try:
            k, rk, b = strip[base + '/kernel'], strip[rk_name], strip[base + '/bias']
        except KeyError as e:
            raise ValueError('%s: GRU layer %r lacks %s' % (path, base, e))
        units = rk.shape[0] if rk.ndim == 2 else -1
        if rk.ndim != 2 or k.ndim != 2 or rk.shape != (units, 3 * units) or k.shape[1] != 3 * units or b.shape != (3 * units,):
            raise ValueError('%s: layer %r has shapes %r %r %r (reset_after GRUs are not supported)'
                             % (path, base, k.shape, rk.shape, b.shape))
        layers.append((k, rk, b))
        used.update({base + '/kernel', rk_name, base + '/bias'})
    units = layers[-1][1].shape[0]
    dense = [k for k in strip if k.endswith('/kernel') and k not in used and strip[k].shape == (units, 1)]
    if len(dense) != 1:
        raise ValueError('%s: expected exactly one Dense(1) kernel of shape (%d, 1), found %r' % (path, units, dense))
    dbase = dense[0][:-len('/kernel')]
    return {'gru': layers, 'dense_kernel': strip[dense[0]],
            'dense_bias': strip.get(dbase + '/bias', np.zeros(1, np.float32)).reshape(-1)}


# ---- encoding -----------------------------------------------------------------------------------
def _tensor_proto(arr):
    arr = np.ascontiguousarray(arr, dtype='<f4')
    shape = b''.join(_enc_field(2, _enc_int(1, int(d))) for d in arr.shape)
    return _enc_int(1, DT_FLOAT) + _enc_field(2, shape) + _enc_field(4, arr.tobytes())


def _attr(key, value_payload):
    return _enc_field(5, _enc_field(1, key.encode()) + _enc_field(2, value_payload))


def _node(name, op, inputs=(), attrs=()):
    body = _enc_field(1, name.encode()) + _enc_field(2, op.encode())
    for i in inputs:
        body += _enc_field(3, i.encode())
    for a in attrs:
        body += a
    return _enc_field(1, body)


def write_frozen_pb(path, weights, dense_name='dense_1'):
    """Frozen GraphDef with the constants of ``weights`` under the names ``precise-convert`` uses."""
    dtype_attr = _attr('dtype', _enc_int(6, DT_FLOAT))
    graph = _node('net_input', 'Placeholder', attrs=[dtype_attr])
    for li, (k, rk, b) in enumerate(weights['gru']):
        base = 'net' if li == 0 else 'net_%d' % (li + 1)
        for var, arr in (('kernel', k), ('recurrent_kernel', rk), ('bias', b)):
            graph += _node('%s/%s' % (base, var), 'Const',
                           attrs=[dtype_attr, _attr('value', _enc_field(8, _tensor_proto(arr)))])
    for var, arr in (('kernel', np.asarray(weights['dense_kernel']).reshape(-1, 1)),
                     ('bias', np.asarray(weights['dense_bias']).reshape(-1))):
        graph += _node('%s/%s' % (dense_name, var), 'Const',
                       attrs=[dtype_attr, _attr('value', _enc_field(8, _tensor_proto(arr)))])
    graph += _node('net_output', 'Identity', inputs=['%s/Sigmoid' % dense_name], attrs=[_attr('T', _enc_int(6, DT_FLOAT))])
    with open(path, 'wb') as f:
        f.write(graph)
