"""Minimal reader of MATLAB ``-v7.3`` MAT-files (HDF5 containers) -- what the Monti loaders need and nothing more.

The reference reads ``raw_data/<dataset>/training_test_dataset.mat`` with h5py (``preprocessing.py:32-55``:
``load_matlab_file`` -- dense datasets transposed on read, sparse ones as groups with ``data`` / ``ir`` / ``jc``).  h5py is not
part of this image, so the subset of the HDF5 file format those files use is read here directly (format specification 1.x /
2.0 of the HDF Group):

* superblock version 0 behind MATLAB's 512-byte user block (addresses relative to the base address);
* "old style" groups: version-1 object headers, symbol-table message, version-1 group B-tree, ``SNOD`` symbol nodes,
  local heap for the link names (nested groups = MATLAB sparse matrices);
* datasets: dataspace (version 1 / 2), fixed-point and IEEE floating-point datatypes, data layout version 3 (contiguous or
  chunked through a version-1 chunk B-tree), filter pipeline with deflate (1) and shuffle (2), header continuation blocks.

Everything else (new-style groups, fractal heaps, variable-length / compound types, external storage, other filters) raises
``NotImplementedError`` naming what was met.  ``File(path)[name]`` returns a numpy array in the file's (row-major) dimension
order -- like ``np.asarray(h5py.File(path)[name])`` -- or a ``Group`` (dict-like) for a group.
"""
import struct
import zlib

import numpy as np

_SIG = b'\x89HDF\r\n\x1a\n'
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Group(dict):
    """Name -> Group or numpy array; a dataset is read (chunks inflated) when it is first asked for."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if callable(v):
            v = v()
            dict.__setitem__(self, k, v)
        return v

    def keys(self):          # (h5py idiom of the reference: ``'ir' in ds.keys()``)
        return dict.keys(self)


class File(object):
    def __init__(self, path):
        with open(path, 'rb') as f:
            self.b = f.read()
        at = -1
        for ub in (0, 512, 1024, 2048, 4096):              # the superblock sits at 0 or behind a power-of-two user block
            if self.b[ub:ub + 8] == _SIG:
                at = ub
                break
        if at < 0:
            raise ValueError('%s: no HDF5 signature (a MATLAB file older than -v7.3?)' % path)
        ver = self.b[at + 8]
        if ver != 0:
            raise NotImplementedError('HDF5 superblock version %d' % ver)
        if self.b[at + 13] != 8 or self.b[at + 14] != 8:
            raise NotImplementedError('HDF5 offsets / lengths of %d / %d bytes' % (self.b[at + 13], self.b[at + 14]))
        self.base = struct.unpack_from('<Q', self.b, at + 24)[0]
        root = at + 56                                       # root group symbol-table entry
        oha, ctype = struct.unpack_from('<QI', self.b, root + 8)
        if ctype == 1:
            btree, heap = struct.unpack_from('<QQ', self.b, root + 24)
            self._root = self._group(btree, heap)
        else:
            self._root = self._object(oha)
        if not isinstance(self._root, Group):
            raise ValueError('root object is not a group')

    # ---- dict-like surface
    def keys(self):
        return self._root.keys()

    def __contains__(self, k):
        return k in self._root

    def __getitem__(self, k):
        return self._root[k]

    # ---- file structure
    def _at(self, addr):
        return self.base + addr

    def _heap_name(self, heap, off):
        h = self._at(heap)
        if self.b[h:h + 4] != b'HEAP':
            raise ValueError('local heap signature missing')
        data = struct.unpack_from('<Q', self.b, h + 24)[0]
        s = self._at(data) + off
        e = self.b.index(b'\x00', s)
        return self.b[s:e].decode('ascii')

    def _group(self, btree, heap):
        g = Group()
        for name_off, oha in self._group_entries(btree):
            g[self._heap_name(heap, name_off)] = self._object(oha)
        return g

    def _group_entries(self, node):
        p = self._at(node)
        if self.b[p:p + 4] == b'SNOD':
            n = struct.unpack_from('<H', self.b, p + 6)[0]
            for i in range(n):
                e = p + 8 + 40 * i
                yield struct.unpack_from('<QQ', self.b, e)
            return
        if self.b[p:p + 4] != b'TREE':
            raise ValueError('group B-tree node signature missing')
        ntype, _level, used = struct.unpack_from('<BBH', self.b, p + 4)
        if ntype != 0:
            raise ValueError('not a group B-tree node')
        q = p + 24 + 8                                       # key 0, then (child, key) pairs
        for _ in range(used):
            child = struct.unpack_from('<Q', self.b, q)[0]
            q += 16
            for ent in self._group_entries(child):
                yield ent

    def _messages(self, oha):
        p = self._at(oha)
        ver = self.b[p]
        if ver != 1:
            raise NotImplementedError('object header version %d' % ver)
        nmsg = struct.unpack_from('<H', self.b, p + 2)[0]
        size = struct.unpack_from('<I', self.b, p + 8)[0]
        blocks = [(p + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            q, left = blocks.pop(0)
            end = q + left
            while q + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from('<HHB', self.b, q)
                body = q + 8
                if mtype == 0x0010:                          # continuation block
                    off, ln = struct.unpack_from('<QQ', self.b, body)
                    blocks.append((self._at(off), ln))
                out.append((mtype, body, msize))
                q = body + msize
        return out

    def _object(self, oha):
        msgs = self._messages(oha)
        kinds = {t: (body, size) for t, body, size in msgs}
        if 0x0011 in kinds:                                  # symbol table: an old-style group
            btree, heap = struct.unpack_from('<QQ', self.b, kinds[0x0011][0])
            return self._group(btree, heap)
        if 0x0008 not in kinds:
            if 0x0002 in kinds or 0x0006 in kinds:
                raise NotImplementedError('new-style group (link messages)')
            raise ValueError('object with neither a data layout nor a symbol table')
        shape = self._dataspace(kinds[0x0001][0])
        dtype = self._datatype(kinds[0x0003][0])
        filters = self._filters(kinds[0x000B][0]) if 0x000B in kinds else []
        layout = kinds[0x0008][0]
        return lambda: self._data(layout, shape, dtype, filters)

    def _dataspace(self, p):
        ver, rank, flags = struct.unpack_from('<BBB', self.b, p)
        if ver == 1:
            q = p + 8
        elif ver == 2:
            q = p + 4
        else:
            raise NotImplementedError('dataspace version %d' % ver)
        return tuple(struct.unpack_from('<%dQ' % rank, self.b, q)) if rank else ()

    def _datatype(self, p):
        cv, b0, _b1, _b2, size = struct.unpack_from('<BBBBI', self.b, p)
        cls = cv & 0x0F
        order = '>' if (b0 & 1) else '<'
        if cls == 0:                                         # fixed point
            signed = (b0 >> 3) & 1
            return np.dtype('%s%s%d' % (order, 'i' if signed else 'u', size))
        if cls == 1:                                         # IEEE floating point
            return np.dtype('%sf%d' % (order, size))
        raise NotImplementedError('HDF5 datatype class %d' % cls)

    def _filters(self, p):
        ver, n = struct.unpack_from('<BB', self.b, p)
        if ver != 1:
            raise NotImplementedError('filter pipeline version %d' % ver)
        q = p + 8
        ids = []
        for _ in range(n):
            fid, nlen, _flags, ncd = struct.unpack_from('<HHHH', self.b, q)
            q += 8 + ((nlen + 7) & ~7) + 4 * ncd
            if ncd & 1:
                q += 4
            ids.append(fid)
        return ids

    def _unfilter(self, raw, filters, mask, itemsize):
        for k in range(len(filters) - 1, -1, -1):            # filters are undone in reverse order
            if mask & (1 << k):
                continue
            fid = filters[k]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                n = len(raw) // itemsize
                raw = np.frombuffer(raw, np.uint8)[:n * itemsize].reshape(itemsize, n).T.tobytes()
            else:
                raise NotImplementedError('HDF5 filter %d' % fid)
        return raw

    def _data(self, p, shape, dtype, filters):
        ver, cls = struct.unpack_from('<BB', self.b, p)
        if ver != 3:
            raise NotImplementedError('data layout version %d' % ver)
        count = int(np.prod(shape)) if shape else 1
        if cls == 0:                                         # compact
            size = struct.unpack_from('<H', self.b, p + 2)[0]
            return np.frombuffer(self.b, dtype, count, p + 4).reshape(shape).copy() if size else np.zeros(shape, dtype)
        if cls == 1:                                         # contiguous
            addr, _size = struct.unpack_from('<QQ', self.b, p + 2)
            if addr == _UNDEF:
                return np.zeros(shape, dtype)
            return np.frombuffer(self.b, dtype, count, self._at(addr)).reshape(shape).copy()
        if cls != 2:
            raise NotImplementedError('data layout class %d' % cls)
        rank1 = self.b[p + 2]                                # dataset rank + 1
        btree = struct.unpack_from('<Q', self.b, p + 3)[0]
        cdims = struct.unpack_from('<%dI' % rank1, self.b, p + 11)
        chunk = tuple(cdims[:-1])
        out = np.zeros(shape, dtype)
        if btree == _UNDEF:
            return out
        for off, mask, addr, size in self._chunks(btree, rank1):
            raw = self._unfilter(self.b[self._at(addr):self._at(addr) + size], filters, mask, dtype.itemsize)
            blk = np.frombuffer(raw, dtype, int(np.prod(chunk))).reshape(chunk)
            sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, chunk, shape))
            out[sel] = blk[tuple(slice(0, s.stop - s.start) for s in sel)]
        return out

    def _chunks(self, node, rank1):
        p = self._at(node)
        if self.b[p:p + 4] != b'TREE':
            raise ValueError('chunk B-tree node signature missing')
        ntype, level, used = struct.unpack_from('<BBH', self.b, p + 4)
        if ntype != 1:
            raise ValueError('not a chunk B-tree node')
        ksz = 8 + 8 * rank1
        q = p + 24
        for _ in range(used):
            size, mask = struct.unpack_from('<II', self.b, q)
            off = struct.unpack_from('<%dQ' % rank1, self.b, q + 8)[:-1]
            child = struct.unpack_from('<Q', self.b, q + ksz)[0]
            q += ksz + 8
            if level == 0:
                yield off, mask, child, size
            else:
                for c in self._chunks(child, rank1):
                    yield c


def load_matlab_field(path, name):
    """The reference's ``load_matlab_file(path_file, name_field)`` (``preprocessing.py:32-55``): a sparse field (group with
    ``data`` / ``ir`` / ``jc``) as ``scipy.sparse.csc_matrix`` float32, a dense one as a float32 array TRANSPOSED (MATLAB is
    column-major)."""
    import scipy.sparse as sp
    ds = File(path)[name]
    if isinstance(ds, Group) and 'ir' in ds.keys():
        return sp.csc_matrix((np.asarray(ds['data']), np.asarray(ds['ir']), np.asarray(ds['jc']))).astype(np.float32)
    return np.asarray(ds).astype(np.float32).T
