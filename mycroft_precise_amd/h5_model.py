"""
Keras ``<model>.net`` files without h5py: a reader for the part of the HDF5 file format that Keras 2.x / h5py write.

The reference stores its trained network with ``model.save('<name>.net')`` (Keras ModelCheckpoint,
/root/reference/precise/scripts/train.py:91-92) and loads it back with ``keras.models.load_model``
(/root/reference/precise/model.py:48-54, network_runner.py:77-95).  The file is HDF5; this module reads the weights
straight from it so that ``Listener('hey-mycroft.net')`` works where neither Keras nor h5py exist (the GPU box).

Scope -- what libhdf5 writes for h5py's default settings (``libver='earliest'``), which is what a Keras ``.net`` is:
superblock version 0 / 1 (and the version 2 / 3 header fields), version-1 object headers with continuation blocks,
old-style groups (symbol-table message -> B-tree v1 -> symbol-table nodes -> local heap), new-style groups only in
their compact form (link messages in a version-2 object header), datasets with contiguous, compact or chunked
(B-tree v1 index; deflate and shuffle filters) layout, little- or big-endian integers and IEEE floats, fixed-length
strings, variable-length strings through the global heap, attributes (message versions 1-3).  Anything else
(dense link / attribute storage in fractal heaps, version-4 chunk indexes, compound or reference types, external
storage) raises ``H5Unsupported`` naming the feature; ``model.load_weights`` then points at the side-car exporter.

Written from the HDF5 File Format Specification (versions 1.1 / 2.0 / 3.0); PINNED ONLY AGAINST ITSELF: no HDF5
library and no real ``.net`` file exist in the build container, so the tests read files produced by
``tests/h5_writer.py`` (same specification, written independently of this reader's code paths: it emits bytes, it
does not share parsing helpers).  Keras layout of a saved Sequential (keras/engine/saving.py, 2.2.4): root attributes
``keras_version``, ``backend``, ``model_config`` (JSON); group ``model_weights`` with attribute ``layer_names``; one
group per layer with attribute ``weight_names``; datasets ``<layer>/<layer>/kernel:0`` ...
"""
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF
# Bounds for damaged / hostile files: every structure is visited once (cyclic B-trees and continuation chains end in
# H5FormatError, not in a hang), and no dataset or attribute may claim more memory than a wake-word model could need
MAX_DATASET_BYTES = 1 << 28         # 256 MiB per dataset (the stock network is 8 KB, the widest kernel holds 2.4 MB)
MAX_HEADER_BLOCKS = 4096            # object-header continuation blocks per object
MAX_TREE_DEPTH = 64


class H5Unsupported(NotImplementedError):
    pass


class H5FormatError(ValueError):
    pass


class _Buf:
    """Little-endian cursor over the file image."""

    def __init__(self, data, pos=0):
        self.d, self.p = data, pos

    def u(self, n):
        if self.p + n > len(self.d):
            raise H5FormatError('truncated file: read of %d bytes at %d' % (n, self.p))
        v = int.from_bytes(self.d[self.p:self.p + n], 'little')
        self.p += n
        return v

    def raw(self, n):
        if self.p + n > len(self.d):
            raise H5FormatError('truncated file: read of %d bytes at %d' % (n, self.p))
        v = self.d[self.p:self.p + n]
        self.p += n
        return v

    def skip(self, n):
        if n < 0:
            raise H5FormatError('negative skip of %d bytes at %d' % (n, self.p))
        self.p += n


class _Datatype:
    def __init__(self, cls, size, order='<', signed=True, strpad=0, base=None, vlen_string=False):
        self.cls, self.size, self.order, self.signed, self.strpad, self.base, self.vlen_string = cls, size, order, signed, strpad, base, vlen_string

    def numpy(self):
        if self.cls == 0:
            return np.dtype('%s%s%d' % (self.order, 'i' if self.signed else 'u', self.size))
        if self.cls == 1:
            if self.size not in (2, 4, 8):
                raise H5Unsupported('%d-byte floating point' % self.size)
            return np.dtype('%sf%d' % (self.order, self.size))
        if self.cls == 3:
            return np.dtype('S%d' % self.size)
        raise H5Unsupported('datatype class %d as an array element' % self.cls)


def _parse_datatype(b, depth=0):
    """Datatype message (0x0003) at the cursor; leaves the cursor behind its properties."""
    if depth > 4:
        raise H5FormatError('datatype nested deeper than 4 levels')
    word = b.u(4)
    cls, version, bits = word & 0xF, (word >> 4) & 0xF, word >> 8
    size = b.u(4)
    if version not in (1, 2, 3):
        raise H5Unsupported('datatype message version %d' % version)
    order = '>' if bits & 1 else '<'
    if cls == 0:                                     # fixed point: bit offset, precision
        b.skip(4)
        if size not in (1, 2, 4, 8):
            raise H5Unsupported('%d-byte integers' % size)
        return _Datatype(0, size, order, signed=bool(bits & 8))
    if cls == 1:                                     # floating point: 12 bytes of layout (IEEE assumed, checked by size)
        if bits & 0x40:
            raise H5Unsupported('VAX byte order')
        b.skip(12)
        return _Datatype(1, size, order)
    if cls == 3:                                     # fixed-length string: padding type in bits 0-3
        if size < 1 or size > MAX_DATASET_BYTES:
            raise H5FormatError('fixed-length string of %d bytes' % size)
        return _Datatype(3, size, strpad=bits & 0xF)
    if cls == 9:                                     # variable length: base type follows
        base = _parse_datatype(b, depth + 1)
        return _Datatype(9, size, base=base, vlen_string=(bits & 0xF) == 1)
    raise H5Unsupported('datatype class %d (compound / reference / enum / array / opaque / time / bitfield)' % cls)


def _parse_dataspace(b, L):
    version, rank, flags = b.u(1), b.u(1), b.u(1)
    if version == 1:
        b.skip(5)
    elif version == 2:
        kind = b.u(1)
        if kind == 2:
            return None                              # null dataspace
    else:
        raise H5Unsupported('dataspace message version %d' % version)
    if rank > 32:
        raise H5FormatError('dataspace of rank %d' % rank)
    dims = tuple(b.u(L) for _ in range(rank))
    if flags & 1:
        b.skip(rank * L)                             # maximum dimensions
    return dims


def _count(shape, itemsize):
    """Elements of a dataspace, refused when they could not be a model's (a flipped bit in a dimension must not
    turn into a multi-gigabyte allocation)."""
    count = 1
    for dim in shape or ():
        count *= int(dim)
        if count * itemsize > MAX_DATASET_BYTES:
            raise H5FormatError('dataspace %r x %d bytes exceeds the %d-byte bound of this reader' % (tuple(shape), itemsize, MAX_DATASET_BYTES))
    return count


class _Object:
    """An object header's messages, parsed on demand into a group or a dataset."""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.msgs = f._read_header(addr)             # [(type, flags, offset of the data, size)]
        self._attrs = None

    def _first(self, kind):
        for m in self.msgs:
            if m[0] == kind:
                return m
        return None

    @property
    def attrs(self):
        if self._attrs is None:
            info = self._first(0x15)                 # attribute info: dense storage in use when its heap address is defined
            if info is not None:
                b = _Buf(self.f.data, info[2])
                b.skip(1)
                if b.u(1) & 1:
                    b.skip(2)
                if b.u(self.f.O) != self.f.undef:
                    raise H5Unsupported('attributes in dense storage (fractal heap)')
            self._attrs = {}
            for kind, _, off, size in self.msgs:
                if kind == 0x0C:
                    name, value = self.f._parse_attribute(off, size)
                    self._attrs[name] = value
        return self._attrs

    # ---- group side ---------------------------------------------------------------------------
    def links(self):
        out = {}
        st = self._first(0x11)
        if st is not None:
            b = _Buf(self.f.data, st[2])
            btree, heap = b.u(self.f.O), b.u(self.f.O)
            out.update(self.f._group_entries(btree, heap))
        for kind, _, off, size in self.msgs:
            if kind == 0x06:
                name, addr = self.f._parse_link(off)
                if addr is not None:
                    out[name] = addr
        li = self._first(0x02)                       # link info: dense storage in use when its heap address is defined
        if li is not None:
            b = _Buf(self.f.data, li[2])
            b.skip(1)
            if b.u(1) & 1:
                b.skip(8)
            if b.u(self.f.O) != self.f.undef:
                raise H5Unsupported('links in dense storage (fractal heap): groups of this size need h5py')
        return out

    def keys(self):
        return list(self.links())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        obj = self
        for part in [p for p in path.split('/') if p]:
            links = obj.links()
            if part not in links:
                raise KeyError(path)
            obj = _Object(self.f, links[part])
        return obj

    # ---- dataset side -------------------------------------------------------------------------
    @property
    def shape(self):
        m = self._first(0x01)
        if m is None:
            raise H5FormatError('object at %d has no dataspace' % self.addr)
        return _parse_dataspace(_Buf(self.f.data, m[2]), self.f.L)

    def _datatype(self):
        m = self._first(0x03)
        if m is None:
            raise H5FormatError('object at %d has no datatype' % self.addr)
        return _parse_datatype(_Buf(self.f.data, m[2]))

    @property
    def dtype(self):
        return self._datatype().numpy()

    def read(self):
        """The dataset as a numpy array in native byte order."""
        f = self.f
        shape = self.shape
        if shape is None:
            return None
        dt = self._datatype()
        nd = dt.numpy()
        count = _count(shape, nd.itemsize)
        lay = self._first(0x08)
        if lay is None:
            raise H5FormatError('dataset at %d has no layout message' % self.addr)
        b = _Buf(f.data, lay[2])
        version = b.u(1)
        if version not in (3, 4):
            raise H5Unsupported('data layout message version %d' % version)
        cls = b.u(1)
        if cls == 0:                                 # compact: the data sits in the header
            n = b.u(2)
            raw = b.raw(n)
        elif cls == 1:                               # contiguous
            addr, n = b.u(f.O), b.u(f.L)
            if addr == f.undef:
                raw = bytes(count * nd.itemsize)     # never written: fill value 0
            else:
                raw = f.data[f.base + addr:f.base + addr + n]
        elif cls == 2:
            if version != 3:
                raise H5Unsupported('version-4 chunk indexes (libver="latest")')
            rank = b.u(1)
            btree = b.u(f.O)
            chunk = tuple(b.u(4) for _ in range(rank))         # last entry: the element size
            if rank != len(shape) + 1 or any(c < 1 for c in chunk):
                raise H5FormatError('dataset at %d: chunk dimensions %r for shape %r' % (self.addr, chunk, shape))
            _count(chunk[:-1], nd.itemsize)
            raw = f._read_chunked(btree, shape, chunk[:-1], nd.itemsize, self._filters())
        else:
            raise H5Unsupported('data layout class %d (virtual)' % cls)
        if len(raw) < count * nd.itemsize:
            raise H5FormatError('dataset at %d: %d bytes of data for %d elements of %d bytes' % (self.addr, len(raw), count, nd.itemsize))
        arr = np.frombuffer(raw, dtype=nd, count=count).reshape(shape)
        return arr.astype(nd.newbyteorder('='))

    def _filters(self):
        m = self._first(0x0B)
        if m is None:
            return []
        b = _Buf(self.f.data, m[2])
        version, n = b.u(1), b.u(1)
        out = []
        if version == 1:
            b.skip(6)
        elif version != 2:
            raise H5Unsupported('filter pipeline version %d' % version)
        for _ in range(n):
            fid = b.u(2)
            name_len = b.u(2) if (version == 1 or fid >= 256) else 0
            b.skip(2)                                # flags
            nvals = b.u(2)
            if version == 1:
                b.skip((name_len + 7) // 8 * 8)
            else:
                b.skip(name_len)
            vals = [b.u(4) for _ in range(nvals)]
            if version == 1 and nvals % 2:
                b.skip(4)
            out.append((fid, vals))
        return out


class H5File(_Object):
    """Read-only view of an HDF5 file: ``f['model_weights/net/net/kernel:0'].read()``, ``f.attrs['model_config']``."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            self.data = fh.read()
        pos = 0
        while self.data[pos:pos + 8] != SIGNATURE:
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(self.data):
                raise H5FormatError('%s is not an HDF5 file (no superblock signature)' % path)
        b = _Buf(self.data, pos + 8)
        version = b.u(1)
        if version in (0, 1):
            b.skip(4)
            self.O, self.L = b.u(1), b.u(1)
            b.skip(1 + 2 + 2 + 4 + (4 if version == 1 else 0))
            self.base = b.u(self.O)
            b.skip(3 * self.O)                       # free space, end of file, driver information
            b.skip(self.O)                           # root entry: link name offset
            root = b.u(self.O)
        elif version in (2, 3):
            self.O, self.L = b.u(1), b.u(1)
            b.skip(1)
            self.base = b.u(self.O)
            b.skip(2 * self.O)
            root = b.u(self.O)
        else:
            raise H5Unsupported('superblock version %d' % version)
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise H5Unsupported('%d-byte offsets / %d-byte lengths' % (self.O, self.L))
        self.undef = UNDEF & ((1 << (8 * self.O)) - 1)
        _Object.__init__(self, self, root)

    # ---- object headers -------------------------------------------------------------------------
    def _read_header(self, addr):
        d = self.data
        a = self.base + addr
        msgs = []
        if d[a:a + 4] == b'OHDR':
            b = _Buf(d, a + 4)
            if b.u(1) != 2:
                raise H5Unsupported('object header version')
            flags = b.u(1)
            if flags & 0x20:
                b.skip(16)
            if flags & 0x10:
                b.skip(4)
            size = b.u(1 << (flags & 3))
            blocks = [(b.p, size)]
            seen = set()
            while blocks:
                start, n = blocks.pop(0)
                if start in seen or len(seen) >= MAX_HEADER_BLOCKS:
                    raise H5FormatError('object header at %d: cyclic or endless continuation chain' % addr)
                seen.add(start)
                b = _Buf(d, start)
                end = start + n
                while b.p + 4 <= end:
                    kind, msize, mflags = b.u(1), b.u(2), b.u(1)
                    if flags & 4:
                        b.skip(2)
                    if kind == 0x10:
                        c = _Buf(d, b.p)
                        caddr, clen = c.u(self.O), c.u(self.L)
                        if d[self.base + caddr:self.base + caddr + 4] != b'OCHK':
                            raise H5FormatError('continuation block without signature at %d' % caddr)
                        if clen < 8:
                            raise H5FormatError('continuation block of %d bytes at %d' % (clen, caddr))
                        blocks.append((self.base + caddr + 4, clen - 8))
                    elif kind != 0:
                        msgs.append((kind, mflags, b.p, msize))
                    b.skip(msize)
            return msgs
        b = _Buf(d, a)
        if b.u(1) != 1:
            raise H5FormatError('no object header at %d' % addr)
        b.skip(1)
        b.skip(2 + 4)                                # number of messages (continuations and nulls included), reference count
        size = b.u(4)
        b.skip(4)                                    # header is padded to 8 bytes
        blocks = [(b.p, size)]
        seen = set()
        while blocks:
            start, n = blocks.pop(0)
            if start in seen or len(seen) >= MAX_HEADER_BLOCKS:
                raise H5FormatError('object header at %d: cyclic or endless continuation chain' % addr)
            seen.add(start)
            b = _Buf(d, start)
            while b.p + 8 <= start + n:
                kind, msize, mflags = b.u(2), b.u(2), b.u(1)
                b.skip(3)
                if mflags & 2:
                    raise H5Unsupported('shared header messages')
                if kind == 0x10:
                    c = _Buf(d, b.p)
                    caddr, clen = c.u(self.O), c.u(self.L)
                    blocks.append((self.base + caddr, clen))
                elif kind != 0:
                    msgs.append((kind, mflags, b.p, msize))
                b.skip(msize)
        return msgs

    # ---- old-style groups -----------------------------------------------------------------------
    def _heap_string(self, heap_data, off):
        start = heap_data + off
        if start < 0 or start >= len(self.data):
            raise H5FormatError('link name at %d lies outside the file' % start)
        end = self.data.find(b'\0', start, start + 4096)
        if end < 0:
            raise H5FormatError('unterminated link name at %d' % start)
        return self.data[start:end].decode('utf-8')

    def _group_entries(self, btree, heap):
        d = self.data
        h = _Buf(d, self.base + heap)
        if h.raw(4) != b'HEAP':
            raise H5FormatError('no local heap at %d' % heap)
        h.skip(4 + 2 * self.L)
        heap_data = self.base + h.u(self.O)
        out = {}
        seen = set()

        def node(addr, depth=0):
            if addr in seen or depth > MAX_TREE_DEPTH:
                raise H5FormatError('group B-tree at %d: cyclic or too deep' % btree)
            seen.add(addr)
            b = _Buf(d, self.base + addr)
            sig = b.raw(4)
            if sig == b'TREE':
                ntype, level, used = b.u(1), b.u(1), b.u(2)
                if ntype != 0:
                    raise H5FormatError('group B-tree node of type %d' % ntype)
                b.skip(2 * self.O)
                for _ in range(used):
                    b.skip(self.L)                   # key
                    node(b.u(self.O), depth + 1)
            elif sig == b'SNOD':
                b.skip(2)
                for _ in range(b.u(2)):
                    name_off, header = b.u(self.O), b.u(self.O)
                    b.skip(24)
                    out[self._heap_string(heap_data, name_off)] = header
            else:
                raise H5FormatError('neither B-tree node nor symbol-table node at %d' % addr)
        node(btree)
        return out

    def _parse_link(self, off):
        b = _Buf(self.data, off)
        if b.u(1) != 1:
            raise H5Unsupported('link message version')
        flags = b.u(1)
        ltype = b.u(1) if flags & 8 else 0
        if flags & 4:
            b.skip(8)
        if flags & 16:
            b.skip(1)
        n = b.u(1 << (flags & 3))
        name = b.raw(n).decode('utf-8')
        if ltype != 0:
            return name, None                        # soft / external links: not followed
        return name, b.u(self.O)

    # ---- attributes -----------------------------------------------------------------------------
    def _parse_attribute(self, off, size):
        b = _Buf(self.data, off)
        version = b.u(1)
        if version not in (1, 2, 3):
            raise H5Unsupported('attribute message version %d' % version)
        flags = b.u(1)
        if version >= 2 and flags & 3:
            raise H5Unsupported('attribute with shared datatype / dataspace')
        n_name, n_dt, n_ds = b.u(2), b.u(2), b.u(2)
        if version == 3:
            b.skip(1)
        pad = (lambda n: (n + 7) // 8 * 8) if version == 1 else (lambda n: n)
        p = b.p
        name = self.data[p:p + n_name].split(b'\0')[0].decode('utf-8')
        p += pad(n_name)
        dt = _parse_datatype(_Buf(self.data, p))
        p += pad(n_dt)
        shape = _parse_dataspace(_Buf(self.data, p), self.L)
        p += pad(n_ds)
        if shape is None:
            return name, None
        count = _count(shape, max(1, dt.size))
        if dt.cls == 9:
            if not dt.vlen_string:
                raise H5Unsupported('variable-length sequence attribute %r' % name)
            b = _Buf(self.data, p)
            vals = []
            for _ in range(count):
                n = b.u(4)
                gaddr, gidx = b.u(self.O), b.u(4)
                vals.append(self._global_heap_object(gaddr, gidx)[:n].decode('utf-8') if n else '')
            value = vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
            return name, value
        nd = dt.numpy()
        if p + count * nd.itemsize > len(self.data):
            raise H5FormatError('attribute %r: %d elements of %d bytes run past the end of the file' % (name, count, nd.itemsize))
        arr = np.frombuffer(self.data[p:p + count * nd.itemsize], dtype=nd, count=count).reshape(shape)
        if dt.cls == 3:
            arr = np.array([v.split(b'\0')[0] if dt.strpad != 2 else v.rstrip(b' ') for v in arr.ravel()], dtype=object).reshape(shape)
            return name, (arr[()] if shape == () else arr)
        arr = arr.astype(nd.newbyteorder('='))
        return name, (arr[()] if shape == () else arr)

    def _global_heap_object(self, addr, index):
        b = _Buf(self.data, self.base + addr)
        if b.raw(4) != b'GCOL':
            raise H5FormatError('no global heap collection at %d' % addr)
        b.skip(4)
        end = self.base + addr + b.u(self.L)
        while b.p + 8 + self.L <= end:
            idx = b.u(2)
            b.skip(6)
            n = b.u(self.L)
            if idx == 0:
                break
            if idx == index:
                return b.raw(n)
            b.skip((n + 7) // 8 * 8)
        raise H5FormatError('global heap object %d not in the collection at %d' % (index, addr))

    # ---- chunked datasets -------------------------------------------------------------------------
    def _read_chunked(self, btree, shape, chunk, itemsize, filters):
        rank = len(shape)
        if btree == self.undef:
            return bytes(int(np.prod(shape, dtype=np.int64)) * itemsize)
        out = np.zeros(shape, dtype='V%d' % itemsize)
        d = self.data
        seen = set()
        chunk_bytes = int(np.prod(chunk, dtype=np.int64)) * itemsize

        def node(addr, depth=0):
            if addr in seen or depth > MAX_TREE_DEPTH:
                raise H5FormatError('chunk B-tree at %d: cyclic or too deep' % btree)
            seen.add(addr)
            b = _Buf(d, self.base + addr)
            if b.raw(4) != b'TREE':
                raise H5FormatError('no chunk B-tree node at %d' % addr)
            ntype, level, used = b.u(1), b.u(1), b.u(2)
            if ntype != 1:
                raise H5FormatError('chunk B-tree node of type %d' % ntype)
            b.skip(2 * self.O)
            for _ in range(used):
                nbytes, mask = b.u(4), b.u(4)
                offs = [b.u(8) for _ in range(rank + 1)][:rank]
                child = b.u(self.O)
                if level > 0:
                    node(child, depth + 1)
                    continue
                raw = d[self.base + child:self.base + child + nbytes]
                for i, (fid, vals) in reversed(list(enumerate(filters))):
                    if mask & (1 << i):
                        continue
                    if fid == 1:
                        z = zlib.decompressobj()
                        raw = z.decompress(raw, chunk_bytes + 64)        # (a chunk inflates to chunk_bytes, + fletcher32)
                        if z.unconsumed_tail:
                            raise H5FormatError('chunk at %d inflates beyond its %d bytes' % (child, chunk_bytes))
                    elif fid == 2:                   # shuffle: byte planes -> elements
                        n = len(raw) // itemsize
                        raw = np.frombuffer(raw[:n * itemsize], np.uint8).reshape(itemsize, n).T.tobytes() + raw[n * itemsize:]
                    elif fid == 3:                   # fletcher32: checksum trails the data
                        raw = raw[:-4]
                    else:
                        raise H5Unsupported('filter %d (only deflate, shuffle, fletcher32)' % fid)
                if len(raw) < chunk_bytes:
                    raise H5FormatError('chunk at %d holds %d of %d bytes' % (child, len(raw), chunk_bytes))
                if any(o >= s_ for o, s_ in zip(offs, shape)):
                    raise H5FormatError('chunk at %d lies outside the dataset (offset %r, shape %r)' % (child, offs, shape))
                block = np.frombuffer(raw, dtype='V%d' % itemsize, count=int(np.prod(chunk, dtype=np.int64))).reshape(chunk)
                sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, shape))
                out[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
        node(btree)
        return out.tobytes()


def _names(value):
    return [v.decode('utf-8') if isinstance(v, (bytes, np.bytes_)) else str(v) for v in np.atleast_1d(value)]


class NotAPreciseModel(ValueError):
    """The file parsed, but it does not hold GRU layer(s) + Dense(1) the way model.py:76-82 builds them."""


def _guard(fn, path):
    """Run the parser; whatever a damaged file makes it trip over (an index past the end, a missing message, a bad
    deflate stream, undecodable text, runaway recursion) surfaces as H5FormatError -- the two documented
    exception types (H5FormatError / H5Unsupported) are the whole error surface of this module."""
    try:
        return fn()
    except (H5Unsupported, H5FormatError, NotAPreciseModel):
        raise
    except (KeyError, IndexError, TypeError, AttributeError, ValueError, OverflowError, zlib.error, UnicodeDecodeError,
            RecursionError, MemoryError) as ex:
        raise H5FormatError('%s: damaged or not a Keras HDF5 file (%s: %s)' % (path, type(ex).__name__, ex)) from ex


def _weights_from_net(path) -> dict:
    f = H5File(path)
    root = f['model_weights'] if 'model_weights' in f else f
    if 'layer_names' not in root.attrs:
        raise H5FormatError('%s: no layer_names attribute: not a Keras model / weights file' % path)
    gru, dense = [], None
    for name in _names(root.attrs['layer_names']):
        g = root[name]
        arrays = {}
        for wname in _names(g.attrs.get('weight_names', [])):
            arrays[wname.split('/')[-1].split(':')[0]] = np.asarray(g[wname].read(), dtype=np.float32)
        if 'recurrent_kernel' in arrays:
            if 'bias' not in arrays or 'kernel' not in arrays:
                raise NotAPreciseModel('%s: GRU layer %s lacks a kernel / bias' % (path, name))
            if arrays['bias'].ndim != 1:
                raise NotAPreciseModel('%s: layer %s is a reset_after GRU (bias %r): not a precise model' % (path, name, arrays['bias'].shape))
            k, rk, bias = arrays['kernel'], arrays['recurrent_kernel'], arrays['bias']
            if k.ndim != 2 or rk.ndim != 2 or rk.shape[1] != 3 * rk.shape[0] or k.shape[1] != rk.shape[1] or bias.shape[0] != rk.shape[1]:
                raise NotAPreciseModel('%s: GRU layer %s has inconsistent shapes %r %r %r' % (path, name, k.shape, rk.shape, bias.shape))
            gru.append((k, rk, bias))
        elif 'kernel' in arrays:
            if arrays['kernel'].ndim != 2:
                raise NotAPreciseModel('%s: Dense layer %s has a kernel of shape %r' % (path, name, arrays['kernel'].shape))
            dense = (arrays['kernel'], arrays.get('bias', np.zeros(arrays['kernel'].shape[1], np.float32)))
    if not gru or dense is None:
        raise NotAPreciseModel('%s: expected GRU layer(s) followed by a Dense(1) layer' % path)
    return {'gru': gru, 'dense_kernel': dense[0], 'dense_bias': dense[1]}


def weights_from_net(path) -> dict:
    """The GRU + Dense weights of a Keras ``.net`` (the layout tools/export_net_to_npz.py documents), as
    ``model.load_weights`` returns them.  model.py:76-82: GRU layer(s) named 'net', then Dense(1).
    Raises H5FormatError (damaged file), H5Unsupported (HDF5 feature outside the subset) or NotAPreciseModel."""
    return _guard(lambda: _weights_from_net(path), path)


def model_config(path):
    """The ``model_config`` JSON of a Keras ``.net`` as a dict, or None (weights-only file)."""
    import json

    def read():
        cfg = H5File(path).attrs.get('model_config')
        if cfg is None:
            return None
        return json.loads(cfg.decode('utf-8') if isinstance(cfg, (bytes, np.bytes_)) else cfg)
    return _guard(read, path)
