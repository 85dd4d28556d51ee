"""
An INDEPENDENT encoder for TensorFlow GraphDef files, for the tests of mycroft_precise_amd/pb_model.py:
the message types are declared from TensorFlow's public .proto field numbers
(tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape,versions}.proto) as runtime
descriptors and serialised by google.protobuf itself -- not by the module under test.

    G = graphdef_classes(packed_floats=True)      # proto3 default: repeated float packed
    G = graphdef_classes(packed_floats=False)     # proto2-style one-tag-per-element float_val

Only the fields a frozen precise model exercises are declared; everything is wire-compatible with the real
schema (same numbers, same types).
"""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_T = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=_T.LABEL_OPTIONAL, type_name=None, packed=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed is not None:
        f.options.packed = packed
    return f


def graphdef_classes(packed_floats=True):
    pkg = 'tfmini_%s' % ('packed' if packed_floats else 'unpacked')
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = pkg + '.proto', pkg, 'proto3'

    shape = fd.message_type.add(); shape.name = 'TensorShapeProto'
    dim = shape.nested_type.add(); dim.name = 'Dim'
    _field(dim, 'size', 1, _T.TYPE_INT64)
    _field(dim, 'name', 2, _T.TYPE_STRING)
    _field(shape, 'dim', 2, _T.TYPE_MESSAGE, _T.LABEL_REPEATED, '.%s.TensorShapeProto.Dim' % pkg)
    _field(shape, 'unknown_rank', 3, _T.TYPE_BOOL)

    tensor = fd.message_type.add(); tensor.name = 'TensorProto'
    _field(tensor, 'dtype', 1, _T.TYPE_INT32)                       # enum DataType on the wire: a varint
    _field(tensor, 'tensor_shape', 2, _T.TYPE_MESSAGE, type_name='.%s.TensorShapeProto' % pkg)
    _field(tensor, 'version_number', 3, _T.TYPE_INT32)
    _field(tensor, 'tensor_content', 4, _T.TYPE_BYTES)
    _field(tensor, 'float_val', 5, _T.TYPE_FLOAT, _T.LABEL_REPEATED, packed=packed_floats)
    _field(tensor, 'int_val', 7, _T.TYPE_INT32, _T.LABEL_REPEATED, packed=True)

    attr = fd.message_type.add(); attr.name = 'AttrValue'
    _field(attr, 's', 2, _T.TYPE_BYTES)
    _field(attr, 'i', 3, _T.TYPE_INT64)
    _field(attr, 'f', 4, _T.TYPE_FLOAT)
    _field(attr, 'b', 5, _T.TYPE_BOOL)
    _field(attr, 'type', 6, _T.TYPE_INT32)
    _field(attr, 'shape', 7, _T.TYPE_MESSAGE, type_name='.%s.TensorShapeProto' % pkg)
    _field(attr, 'tensor', 8, _T.TYPE_MESSAGE, type_name='.%s.TensorProto' % pkg)

    node = fd.message_type.add(); node.name = 'NodeDef'
    entry = node.nested_type.add(); entry.name = 'AttrEntry'; entry.options.map_entry = True
    _field(entry, 'key', 1, _T.TYPE_STRING)
    _field(entry, 'value', 2, _T.TYPE_MESSAGE, type_name='.%s.AttrValue' % pkg)
    _field(node, 'name', 1, _T.TYPE_STRING)
    _field(node, 'op', 2, _T.TYPE_STRING)
    _field(node, 'input', 3, _T.TYPE_STRING, _T.LABEL_REPEATED)
    _field(node, 'device', 4, _T.TYPE_STRING)
    _field(node, 'attr', 5, _T.TYPE_MESSAGE, _T.LABEL_REPEATED, '.%s.NodeDef.AttrEntry' % pkg)

    ver = fd.message_type.add(); ver.name = 'VersionDef'
    _field(ver, 'producer', 1, _T.TYPE_INT32)
    _field(ver, 'min_consumer', 2, _T.TYPE_INT32)

    graph = fd.message_type.add(); graph.name = 'GraphDef'
    _field(graph, 'node', 1, _T.TYPE_MESSAGE, _T.LABEL_REPEATED, '.%s.NodeDef' % pkg)
    _field(graph, 'version', 3, _T.TYPE_INT32)
    _field(graph, 'versions', 4, _T.TYPE_MESSAGE, type_name='.%s.VersionDef' % pkg)

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, 'GetMessageClass', None)
    if get is None:                                                   # older protobuf
        factory = message_factory.MessageFactory(pool)
        get = factory.GetPrototype

    class G:
        pass
    for name in ('TensorShapeProto', 'TensorProto', 'AttrValue', 'NodeDef', 'VersionDef', 'GraphDef'):
        setattr(G, name, get(pool.FindMessageTypeByName('%s.%s' % (pkg, name))))
    return G


DT_FLOAT, DT_INT32 = 1, 3


def add_const(G, graph, name, array, encoding='content', prefix=''):
    """A float32 Const node the way ``convert_variables_to_constants`` leaves it.  encoding: 'content'
    (tensor_content bytes, what TF writes for real weights), 'float_val' (one value per element) or 'splat'
    (a single float_val for a constant-filled tensor)."""
    import numpy as np
    array = np.asarray(array, dtype='<f4')
    n = graph.node.add()
    n.name, n.op = prefix + name, 'Const'
    n.attr['dtype'].type = DT_FLOAT
    t = n.attr['value'].tensor
    t.dtype = DT_FLOAT
    for d in array.shape:
        t.tensor_shape.dim.add().size = int(d)
    if encoding == 'content':
        t.tensor_content = array.tobytes()
    elif encoding == 'float_val':
        t.float_val.extend(float(v) for v in array.reshape(-1))
    elif encoding == 'splat':
        t.float_val.append(float(array.reshape(-1)[0]))
    else:
        raise ValueError(encoding)
    return n
