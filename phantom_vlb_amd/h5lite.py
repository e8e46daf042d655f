"""Read-only HDF5 subset in pure Python / numpy, for the two file kinds the reference produces.

The reference's data files are HDF5 written by h5py with default settings:
  * lazy-load sample stores (``src/preprocessing/videollama2_vlb_lazyloading.py:141-164``): groups ``"{i}"`` holding
    six contiguous datasets each + a root dataset ``dset_len``; read by ``VLB_Dataset`` (``src/datamodule/...:83-109``);
  * per-episode feature / BOLD files (``src/preprocessing/videollama2_vlb_extractfeatures.py:443-508``): groups of
    gzip-4 chunked datasets.
h5py is not installed where this package is built and benchmarked, so ``VLB_Dataset`` and ``episodes`` fall back to this
reader when ``import h5py`` fails.  Supported (HDF5 file format spec, the part libhdf5 1.8-1.14 emits with
``libver='earliest'``, h5py's default): superblock v0-v3, object headers v1 and v2 with continuation blocks, old-style
groups (symbol table: B-tree v1 + local heap) and compact new-style groups (link messages), simple dataspaces, fixed-point
and IEEE float datatypes of either byte order, contiguous / compact / chunked (B-tree v1) layouts, the deflate and shuffle
filters.  Anything else raises ``NotImplementedError`` naming the feature - it never guesses.

Pinned by ``tests/test_cpu_h5lite.py`` against fixtures written by the real h5py 3.3 / libhdf5 1.10.6
(``tests/golden/make_h5_fixtures.py``).
"""
from __future__ import annotations

import zlib

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"


class H5Error(NotImplementedError):
    pass


class _Reader:
    def __init__(self, path):
        self.buf = np.memmap(path, dtype=np.uint8, mode="r")
        n = self.buf.shape[0]
        base = 0
        while base < n and bytes(self.buf[base:base + 8]) != SIG:
            base = 512 if base == 0 else base * 2
        if base >= n:
            raise ValueError(f"{path}: not an HDF5 file")
        self.sb = base
        self.base_addr = 0
        ver = int(self.buf[base + 8])
        if ver in (0, 1):
            self.O, self.L = int(self.buf[base + 13]), int(self.buf[base + 14])
            p = base + 24 + (4 if ver == 1 else 0)
            self.base_addr = self.uint(p, self.O)
            p += 4 * self.O                           # base, free-space, eof, driver
            self.root_header = self.addr(p + self.O)  # root symbol table entry: name offset, header address
        elif ver in (2, 3):
            self.O, self.L = int(self.buf[base + 9]), int(self.buf[base + 10])
            p = base + 12
            self.base_addr = self.uint(p, self.O)
            self.root_header = self.addr(p + 3 * self.O)
        else:
            raise H5Error(f"HDF5 superblock version {ver}")

    # ---- primitives
    def raw(self, off, n):
        return bytes(self.buf[off:off + n])

    def uint(self, off, n):
        return int.from_bytes(self.raw(off, n), "little")

    def addr(self, off):
        v = self.uint(off, self.O)
        return None if v == (1 << (8 * self.O)) - 1 else v + self.base_addr

    def length(self, off):
        return self.uint(off, self.L)

    # ---- object headers -> list of (type, flags, payload offset, size)
    def messages(self, hdr):
        out = []
        if self.raw(hdr, 4) == b"OHDR":
            ver, flags = self.buf[hdr + 4], int(self.buf[hdr + 5])
            if ver != 2:
                raise H5Error(f"object header version {ver}")
            p = hdr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szb = 1 << (flags & 3)
            chunk = self.uint(p, szb)
            p += szb
            blocks = [(p, chunk)]
            while blocks:
                q, size = blocks.pop(0)
                end = q + size
                while q + 4 <= end:
                    t, s, f = int(self.buf[q]), self.uint(q + 1, 2), int(self.buf[q + 3])
                    q += 4 + (2 if flags & 0x04 else 0)
                    if t == 0x10:
                        a, ln = self.addr(q), self.length(q + self.O)
                        if self.raw(a, 4) != b"OCHK":
                            raise ValueError("bad object header continuation")
                        blocks.append((a + 4, ln - 8))
                    elif t != 0:
                        out.append((t, f, q, s))
                    q += s
            return out
        ver = int(self.buf[hdr])
        if ver != 1:
            raise H5Error(f"object header version {ver}")
        nmsg, size = self.uint(hdr + 2, 2), self.uint(hdr + 8, 4)
        blocks = [(hdr + 16, size)]
        while blocks and len(out) < nmsg + 64:
            q, size = blocks.pop(0)
            end = q + size
            while q + 8 <= end:
                t, s, f = self.uint(q, 2), self.uint(q + 2, 2), int(self.buf[q + 4])
                q += 8
                if t == 0x10:
                    blocks.append((self.addr(q), self.length(q + self.O)))
                elif t != 0:
                    out.append((t, f, q, s))
                q += s
        return out

    # ---- groups
    def links(self, hdr):
        """{name: object header address} of a group."""
        out = {}
        for t, _, p, s in self.messages(hdr):
            if t == 0x11:                                     # symbol table: B-tree v1 + local heap
                btree, heap = self.addr(p), self.addr(p + self.O)
                if self.raw(heap, 4) != b"HEAP":
                    raise ValueError("bad local heap")
                data = self.addr(heap + 8 + 2 * self.L)
                self._walk_group_btree(btree, data, out)
            elif t == 0x06:                                   # link message (compact new-style group)
                ver, fl = int(self.buf[p]), int(self.buf[p + 1])
                q = p + 2
                ltype = 0
                if fl & 0x08:
                    ltype = int(self.buf[q]); q += 1
                if fl & 0x04:
                    q += 8
                if fl & 0x10:
                    q += 1
                nb = 1 << (fl & 3)
                ln = self.uint(q, nb); q += nb
                name = self.raw(q, ln).decode("utf-8"); q += ln
                if ltype != 0:
                    raise H5Error("soft / external links")
                out[name] = self.addr(q)
            elif t == 0x02:                                   # link info: dense storage (fractal heap) unsupported
                fl = int(self.buf[p + 1])
                q = p + 2 + (8 if fl & 1 else 0)
                if self.addr(q) is not None:
                    raise H5Error("dense link storage (fractal heap)")
        return out

    def _walk_group_btree(self, node, heap_data, out):
        sig = self.raw(node, 4)
        if sig == b"SNOD":
            n = self.uint(node + 6, 2)
            q = node + 8
            for _ in range(n):
                name_off, hdr = self.uint(q, self.O), self.addr(q + self.O)
                end = heap_data + name_off
                while self.buf[end] != 0:
                    end += 1
                out[self.raw(heap_data + name_off, end - heap_data - name_off).decode("utf-8")] = hdr
                q += 2 * self.O + 24
            return
        if sig != b"TREE":
            raise ValueError("bad group B-tree node")
        if self.buf[node + 4] != 0:
            raise ValueError("not a group B-tree")
        used = self.uint(node + 6, 2)
        q = node + 8 + 2 * self.O
        for _ in range(used):
            q += self.L                                       # key
            self._walk_group_btree(self.addr(q), heap_data, out)
            q += self.O
        return

    # ---- datasets
    def dataset(self, hdr):
        shape = dtype = layout = None
        filters = []
        for t, _, p, s in self.messages(hdr):
            if t == 0x01:
                ver, rank, fl = int(self.buf[p]), int(self.buf[p + 1]), int(self.buf[p + 2])
                q = p + (8 if ver == 1 else 4)
                if ver not in (1, 2):
                    raise H5Error(f"dataspace version {ver}")
                shape = tuple(self.length(q + i * self.L) for i in range(rank))
            elif t == 0x03:
                cls, bits0 = int(self.buf[p]) & 0x0f, int(self.buf[p + 1])
                size = self.uint(p + 4, 4)
                order = ">" if bits0 & 1 else "<"
                if cls == 0:
                    dtype = np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}")
                elif cls == 1:
                    dtype = np.dtype(f"{order}f{size}")
                else:
                    raise H5Error(f"datatype class {cls}")
            elif t == 0x08:
                ver, cls = int(self.buf[p]), int(self.buf[p + 1])
                if ver != 3:
                    raise H5Error(f"data layout version {ver}")
                if cls == 0:
                    n = self.uint(p + 2, 2)
                    layout = ("compact", p + 4, n)
                elif cls == 1:
                    layout = ("contiguous", self.addr(p + 2), self.length(p + 2 + self.O))
                elif cls == 2:
                    nd = int(self.buf[p + 2])
                    bt = self.addr(p + 3)
                    dims = tuple(self.uint(p + 3 + self.O + 4 * i, 4) for i in range(nd))
                    layout = ("chunked", bt, dims)
                else:
                    raise H5Error(f"layout class {cls}")
            elif t == 0x0B:
                ver, nf = int(self.buf[p]), int(self.buf[p + 1])
                q = p + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid = self.uint(q, 2)
                    if ver == 1 or fid >= 256:
                        nlen = self.uint(q + 2, 2); q += 4
                    else:
                        nlen = 0; q += 2
                    ncd = self.uint(q + 2, 2)
                    q += 4
                    q += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                    cd = [self.uint(q + 4 * i, 4) for i in range(ncd)]
                    q += 4 * ncd + (4 if ver == 1 and ncd % 2 else 0)
                    filters.append((fid, cd))
        if shape is None or dtype is None or layout is None:
            raise ValueError("object is not a simple dataset")
        return shape, dtype, layout, filters

    def read(self, hdr):
        shape, dtype, layout, filters = self.dataset(hdr)
        count = int(np.prod(shape)) if shape else 1
        if layout[0] == "contiguous":
            if layout[1] is None:
                return np.zeros(shape, dtype.newbyteorder("="))
            a = np.frombuffer(self.buf, dtype=dtype, count=count, offset=layout[1])
        elif layout[0] == "compact":
            a = np.frombuffer(self.buf, dtype=dtype, count=count, offset=layout[1])
        else:
            a = self._read_chunked(shape, dtype, layout, filters)
        return np.array(a.reshape(shape)).astype(dtype.newbyteorder("="), copy=False)

    def _read_chunked(self, shape, dtype, layout, filters):
        _, btree, cdims = layout
        rank = len(shape)
        chunk_shape = cdims[:rank]
        out = np.zeros(shape, dtype)
        if btree is None:
            return out
        for f, _ in filters:
            if f not in (1, 2):
                raise H5Error(f"filter id {f} (only deflate and shuffle are supported)")

        def visit(node):
            if self.raw(node, 4) != b"TREE" or self.buf[node + 4] != 1:
                raise ValueError("bad chunk B-tree node")
            level, used = int(self.buf[node + 5]), self.uint(node + 6, 2)
            q = node + 8 + 2 * self.O
            for _ in range(used):
                csize, mask = self.uint(q, 4), self.uint(q + 4, 4)
                offs = [self.uint(q + 8 + 8 * i, 8) for i in range(rank + 1)]
                q += 8 + 8 * (rank + 1)
                child = self.addr(q)
                q += self.O
                if level > 0:
                    visit(child)
                    continue
                data = self.raw(child, csize)
                for i in range(len(filters) - 1, -1, -1):       # undo the pipeline, last filter first
                    if mask & (1 << i):
                        continue
                    fid = filters[i][0]
                    if fid == 1:
                        data = zlib.decompress(data)
                    else:                                       # shuffle: bytes of every element were de-interleaved
                        es = dtype.itemsize
                        arr = np.frombuffer(data, np.uint8)
                        n = arr.size // es
                        data = arr[:n * es].reshape(es, n).T.tobytes() + arr[n * es:].tobytes()
                chunk = np.frombuffer(data, dtype=dtype, count=int(np.prod(chunk_shape))).reshape(chunk_shape)
                sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk_shape, shape))
                sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
                out[sl_out] = chunk[sl_in]
        visit(btree)
        return out


class Dataset:
    def __init__(self, reader, hdr):
        self._r, self._h = reader, hdr
        self.shape, self.dtype, _, _ = reader.dataset(hdr)

    def __array__(self, dtype=None, copy=None):
        a = self._r.read(self._h)
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, key):
        return self._r.read(self._h)[key]

    def __len__(self):
        return self.shape[0]


class Group:
    def __init__(self, reader, hdr):
        self._r, self._h = reader, hdr
        self._links = None

    def _l(self):
        if self._links is None:
            self._links = self._r.links(self._h)
        return self._links

    def keys(self):
        return self._l().keys()

    def __contains__(self, name):
        return name in self._l()

    def __iter__(self):
        return iter(self._l())

    def items(self):
        return ((k, self[k]) for k in self._l())

    def __getitem__(self, name):
        node = self
        for part in [p for p in name.split("/") if p]:
            hdr = node._l().get(part)
            if hdr is None:
                raise KeyError(name)
            types = {t for t, _, _, _ in node._r.messages(hdr)}
            node = Dataset(node._r, hdr) if 0x08 in types else Group(node._r, hdr)
        return node


class File(Group):
    """``h5lite.File(path)`` - the read-only slice of ``h5py.File(path, "r")`` this package uses: ``f["a"]["b"]``,
    ``f["a/b"]``, ``keys()``, ``items()``, ``np.array(dataset)``, ``dataset[...]``, ``.shape`` / ``.dtype``."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise H5Error("h5lite is read-only")
        r = _Reader(path)
        super().__init__(r, r.root_header)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
