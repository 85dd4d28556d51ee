"""
Test helper: writes small HDF5 files byte by byte in the dialect libhdf5 produces for h5py's defaults (superblock
version 0, version-1 object headers, symbol-table groups, B-tree v1), following the HDF5 File Format Specification.
No HDF5 library exists in the build container, so this is what ``mycroft_precise_amd.h5_model`` is tested against; it
shares no code with the reader (it only ever emits bytes).

    tree = group({'model_weights': group({...}, attrs={'layer_names': [b'net', b'dense_1']})}, attrs={...})
    write_h5(path, tree)

Nodes: ``group(children, attrs)``; ``dataset(array, attrs, layout='contiguous' | 'compact' | 'chunked', chunks=None,
gzip=False, shuffle=False)``.  Attribute values: bytes (fixed-length string), str (variable-length string through the
global heap), list of bytes (array of fixed-length strings), numpy arrays / numbers.
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16


def group(children=None, attrs=None):
    return {'kind': 'group', 'children': dict(children or {}), 'attrs': dict(attrs or {})}


def dataset(array, attrs=None, layout='contiguous', chunks=None, gzip=False, shuffle=False):
    return {'kind': 'dataset', 'array': np.asarray(array), 'attrs': dict(attrs or {}), 'layout': layout, 'chunks': chunks,
            'gzip': gzip, 'shuffle': shuffle}


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


class _Writer:
    def __init__(self, split_headers=False, latest=False):
        self.buf = bytearray(96)                 # the superblock goes here at the end
        self.split = split_headers               # put the attributes of every object into a continuation block
        self.latest = latest                     # libver='latest' structures: version-2 object headers, link messages
        self.gheap = []                          # variable-length strings: objects of one global heap collection
        self.gheap_addr = None

    def alloc(self, data, align=8):
        self.buf += b'\0' * (-len(self.buf) % align)
        addr = len(self.buf)
        self.buf += data
        return addr

    # ---- messages ---------------------------------------------------------------------------
    @staticmethod
    def datatype(dt):
        dt = np.dtype(dt)
        if dt.kind == 'f':
            order = 1 if dt.byteorder == '>' else 0
            if dt.itemsize == 4:
                return struct.pack('<BBBBI', 0x11, 0x20 | order, 31, 0, 4) + struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)
            if dt.itemsize == 8:
                return struct.pack('<BBBBI', 0x11, 0x20 | order, 63, 0, 8) + struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)
        if dt.kind in 'iu':
            order = 1 if dt.byteorder == '>' else 0
            return struct.pack('<BBBBI', 0x10, order | (8 if dt.kind == 'i' else 0), 0, 0, dt.itemsize) + struct.pack('<HH', 0, 8 * dt.itemsize)
        if dt.kind == 'S':
            return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, dt.itemsize)           # null-padded, ASCII
        raise ValueError(dt)

    @staticmethod
    def vlen_string_type():
        base = struct.pack('<BBBBI', 0x13, 0x00, 0, 0, 1)
        return struct.pack('<BBBBI', 0x19, 0x01, 0x01, 0, 16) + base               # variable-length string, UTF-8

    @staticmethod
    def dataspace(shape):
        return struct.pack('<BBBB4x', 1, len(shape), 0, 0) + b''.join(struct.pack('<Q', int(d)) for d in shape)

    def attribute(self, name, value):
        nm = name.encode() + b'\0'
        if isinstance(value, str):                                # variable-length string: (length, heap address, index)
            raw = value.encode('utf-8')
            dt, ds = self.vlen_string_type(), self.dataspace(())
            data = struct.pack('<IQI', len(raw), self.gheap_addr, self.gheap.index(raw) + 1)
        else:
            if isinstance(value, bytes):
                arr = np.array(value, dtype='S%d' % max(1, len(value)))
            elif isinstance(value, (list, tuple)) and value and isinstance(value[0], bytes):
                arr = np.array(value, dtype='S%d' % max(len(v) for v in value))
            else:
                arr = np.asarray(value)
            dt, ds, data = self.datatype(arr.dtype), self.dataspace(arr.shape), arr.tobytes()
        if self.latest:                                           # version 3: no padding, a character-set byte
            return struct.pack('<BBHHHB', 3, 0, len(nm), len(dt), len(ds), 0) + nm + dt + ds + data
        return struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data

    def global_heap(self, tree):
        """One global heap collection with every variable-length string of the tree, written first so that the
        attributes can point at it."""
        def walk(node):
            for v in node['attrs'].values():
                if isinstance(v, str) and v.encode('utf-8') not in self.gheap:
                    self.gheap.append(v.encode('utf-8'))
            for c in node.get('children', {}).values():
                walk(c)
        walk(tree)
        if not self.gheap:
            return
        objs = b''
        for i, raw in enumerate(self.gheap):
            objs += struct.pack('<HH4xQ', i + 1, 1, len(raw)) + _pad8(raw)
        size = max(4096, 16 + len(objs) + 16)
        free = size - 16 - len(objs)
        self.gheap_addr = self.alloc(b'GCOL' + struct.pack('<B3xQ', 1, size) + objs + struct.pack('<HH4xQ', 0, 0, free) + b'\0' * (free - 16))

    def header(self, messages, attrs):
        """Version-1 object header from (type, data) messages, the attribute messages (0x0C) last -- with
        ``split_headers`` in a continuation block of their own."""
        def pack(msgs):
            out = b''
            for kind, data in msgs:
                data = _pad8(data)
                out += struct.pack('<HHB3x', kind, len(data), 0) + data
            return out
        amsgs = [(0x0C, self.attribute(name, value)) for name, value in attrs.items()]
        messages = list(messages)
        if self.latest:
            # "OHDR", version 2, flags 0x02 (4-byte chunk size), messages with 4-byte prefixes and no padding, checksum
            # (written as 0: the reader under test does not verify checksums); continuation blocks carry "OCHK"
            def pack2(msgs):
                return b''.join(struct.pack('<BHB', kind, len(data), 0) + data for kind, data in msgs)
            if self.split and amsgs:
                block = b'OCHK' + pack2(amsgs) + b'\0' * 4
                cont = self.alloc(block)
                body = pack2(messages + [(0x10, struct.pack('<QQ', cont, len(block)))])
            else:
                body = pack2(messages + amsgs)
            return self.alloc(b'OHDR' + struct.pack('<BBI', 2, 0x02, len(body)) + body + b'\0' * 4)
        if self.split and amsgs:
            block = pack(amsgs)
            cont = self.alloc(block)
            body = pack(messages + [(0x10, struct.pack('<QQ', cont, len(block)))])
            n = len(messages) + 1 + len(amsgs)
        else:
            body = pack(messages + amsgs)
            n = len(messages) + len(amsgs)
        return self.alloc(struct.pack('<BBHII4x', 1, 0, n, 1, len(body)) + body)

    # ---- objects ------------------------------------------------------------------------------
    def write_dataset(self, node):
        arr = node['array']
        if arr.dtype.kind == 'f' and arr.dtype.byteorder == '=':
            arr = arr.astype(arr.dtype.newbyteorder('<'))
        raw = arr.tobytes()
        msgs = [(0x01, self.dataspace(arr.shape)), (0x03, self.datatype(arr.dtype)), (0x05, struct.pack('<BBBB', 2, 2, 2, 0))]
        if node['layout'] == 'compact':
            msgs.append((0x08, struct.pack('<BBH', 3, 0, len(raw)) + raw))
        elif node['layout'] == 'contiguous':
            addr = self.alloc(raw) if raw else UNDEF
            msgs.append((0x08, struct.pack('<BBQQ', 3, 1, addr, len(raw))))
        else:
            chunks = tuple(node['chunks'] or arr.shape)
            rank, item = arr.ndim, arr.dtype.itemsize
            filters = ([2] if node['shuffle'] else []) + ([1] if node['gzip'] else [])
            entries = []
            grid = [range(0, s, c) for s, c in zip(arr.shape, chunks)]
            for offs in np.array(np.meshgrid(*grid, indexing='ij')).reshape(rank, -1).T if rank else [()]:
                block = np.zeros(chunks, dtype=arr.dtype)
                sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunks, arr.shape))
                block[tuple(slice(0, s.stop - s.start) for s in sel)] = arr[sel]
                data = block.tobytes()
                if node['shuffle']:
                    data = np.frombuffer(data, np.uint8).reshape(-1, item).T.tobytes()
                if node['gzip']:
                    data = zlib.compress(data, 4)
                entries.append((len(data), tuple(int(o) for o in offs), self.alloc(data)))
            node_bytes = b'TREE' + struct.pack('<BBHQQ', 1, 0, len(entries), UNDEF, UNDEF)
            for n, offs, caddr in entries:
                node_bytes += struct.pack('<II', n, 0) + b''.join(struct.pack('<Q', o) for o in offs) + struct.pack('<Q', 0) + struct.pack('<Q', caddr)
            node_bytes += struct.pack('<II', 0, 0) + b''.join(struct.pack('<Q', s) for s in arr.shape) + struct.pack('<Q', 0)
            btree = self.alloc(node_bytes)
            msgs.append((0x08, struct.pack('<BBBQ', 3, 2, rank + 1, btree) + b''.join(struct.pack('<I', c) for c in chunks) + struct.pack('<I', item)))
            if filters:
                pipe = struct.pack('<BB6x', 1, len(filters))
                for fid in filters:
                    vals = [4] if fid == 1 else [item]
                    pipe += struct.pack('<HHHH', fid, 0, 1, len(vals)) + b''.join(struct.pack('<I', v) for v in vals) + b'\0' * 4
                msgs.append((0x0B, pipe))
        return self.header(msgs, node['attrs'])

    def write_group(self, node):
        """Returns (object header address, B-tree address, local heap address)."""
        if self.latest:
            # compact new-style group: link info (no dense storage), group info, one link message per child
            msgs = [(0x02, struct.pack('<BBQQ', 0, 0, UNDEF, UNDEF)), (0x0A, struct.pack('<BB', 0, 0))]
            for name in sorted(node['children']):
                child = node['children'][name]
                addr = self.write_group(child)[0] if child['kind'] == 'group' else self.write_dataset(child)
                nm = name.encode('utf-8')
                msgs.append((0x06, struct.pack('<BBBB', 1, 0x10, 1, len(nm)) + nm + struct.pack('<Q', addr)))      # flags: character set present (UTF-8)
            return self.header(msgs, node['attrs']), UNDEF, UNDEF
        entries = []
        for name in sorted(node['children']):
            child = node['children'][name]
            if child['kind'] == 'group':
                addr, bt, hp = self.write_group(child)
                entries.append((name, addr, 1, struct.pack('<QQ', bt, hp)))
            else:
                entries.append((name, self.write_dataset(child), 0, b'\0' * 16))
        heap = bytearray(8)                          # offset 0: the empty string
        offs = []
        for name, _, _, _ in entries:
            offs.append(len(heap))
            heap += _pad8(name.encode() + b'\0')
        heap += b'\0' * 16
        heap_data = self.alloc(bytes(heap))
        heap_addr = self.alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), UNDEF, heap_data))
        snods = []
        for i in range(0, max(1, len(entries)), 2 * LEAF_K):
            part = list(zip(entries[i:i + 2 * LEAF_K], offs[i:i + 2 * LEAF_K]))
            body = b'SNOD' + struct.pack('<BBH', 1, 0, len(part))
            for (name, addr, cache, scratch), off in part:
                body += struct.pack('<QQII', off, addr, cache, 0) + scratch
            body += b'\0' * (40 * (2 * LEAF_K - len(part)))
            snods.append((self.alloc(body), part[-1][1] if part else 0))
        tree = b'TREE' + struct.pack('<BBHQQ', 0, 0, len(snods), UNDEF, UNDEF) + struct.pack('<Q', 0)
        for addr, last_off in snods:
            tree += struct.pack('<QQ', addr, last_off)
        tree += b'\0' * (16 * (2 * INTERNAL_K - len(snods)))
        btree = self.alloc(tree)
        header = self.header([(0x11, struct.pack('<QQ', btree, heap_addr))], node['attrs'])
        return header, btree, heap_addr


def write_h5(path, tree, split_headers=False, superblock_at=0, latest=False):
    """``split_headers``: attributes go into object-header continuation blocks.  ``superblock_at``: 0 or 512 (a user
    block in front of the file; addresses are relative to the superblock's base address then).  ``latest``: the structures
    of libver='latest' files (superblock 2, version-2 object headers, link messages, version-3 attributes)."""
    w = _Writer(split_headers, latest)
    w.global_heap(tree)
    root, btree, heap = w.write_group(tree)
    eof = len(w.buf)
    if latest:
        sb = b'\x89HDF\r\n\x1a\n' + struct.pack('<BBBB', 2, 8, 8, 0) + struct.pack('<QQQQ', superblock_at, UNDEF, eof, root) + b'\0' * 4
        w.buf[0:len(sb)] = sb
        with open(path, 'wb') as f:
            f.write(b'\0' * superblock_at + bytes(w.buf))
        return
    sb = b'\x89HDF\r\n\x1a\n' + struct.pack('<BBBBBBBB', 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack('<HHI', LEAF_K, INTERNAL_K, 0)
    sb += struct.pack('<QQQQ', superblock_at, UNDEF, eof, UNDEF)
    sb += struct.pack('<QQII', 0, root, 1, 0) + struct.pack('<QQ', btree, heap)
    assert len(sb) == 96
    w.buf[0:96] = sb
    with open(path, 'wb') as f:
        f.write(b'\0' * superblock_at + bytes(w.buf))
