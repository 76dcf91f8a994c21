"""A small read-only HDF5 reader (numpy + zlib, no h5py / libhdf5): enough of the file format to load the reference's point-cloud
files -- `h5py.File(path, 'r')['poisson_<np>'][:]` (Generation/H5DataLoader.py:14-17) -- where h5py is not installed.

Supported (what h5py / libhdf5 write for `create_dataset(name, data=array[, chunks=..., compression='gzip', shuffle=True])`):
  * superblock versions 0-3; groups as symbol tables (B-tree v1 + local heap: h5py's default `libver='earliest'`) or as compact
    link messages (object header v2, `libver='latest'`); nested groups through "a/b/c" paths;
  * datasets of IEEE float32/float64 and 1/2/4/8-byte integers, little or big endian, any rank;
  * layouts: compact, contiguous, chunked with the version-1 B-tree chunk index (data-layout message v3) and, for latest-format
    files, the single-chunk / implicit / fixed-array indexes (data-layout message v4);
  * filters: deflate (gzip), shuffle, fletcher32 (checksum stripped, not verified).
Anything else (dense link storage, extensible-array / B-tree-v2 chunk indexes, compound / string / variable-length types, external
storage, SZIP/LZF) raises H5Error with the feature's name.

The layout follows "HDF5 File Format Specification Version 3.0".  Checked in tests/test_h5_reader.py against files written by the
real HDF5 library (tests/golden/make_h5_fixtures.py drives libhdf5 1.10.6 through ctypes)."""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(RuntimeError):
    pass


class File:
    """`File(path)[name]` -> numpy array (name may be a path "group/dataset"); `keys(group="/")` lists a group; `name in file`."""

    def __init__(self, path: str):
        with open(path, "rb") as f:
            self.buf = f.read()
        self.path = path
        self._superblock()

    # ------------------------------------------------------------------ primitives
    def _u(self, off: int, n: int) -> int:
        return int.from_bytes(self.buf[off:off + n], "little")

    def _addr(self, off: int) -> int:
        v = self._u(off, self.so)
        return UNDEF if v == (1 << (8 * self.so)) - 1 else v + self.base

    def _len(self, off: int) -> int:
        return self._u(off, self.sl)

    # ------------------------------------------------------------------ superblock
    def _superblock(self):
        sig = b"\x89HDF\r\n\x1a\n"
        off = 0
        while self.buf[off:off + 8] != sig:
            off = 512 if off == 0 else off * 2
            if off + 8 > len(self.buf):
                raise H5Error("%s: not an HDF5 file (no superblock signature)" % self.path)
        ver = self.buf[off + 8]
        self.base = 0
        if ver in (0, 1):
            self.so, self.sl = self.buf[off + 13], self.buf[off + 14]
            p = off + 24 + (4 if ver == 1 else 0)
            self.base = self._u(p, self.so)
            p += 4 * self.so                                     # base, free-space, end-of-file, driver-info addresses
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root = self._u(p + self.so, self.so) + self.base
        elif ver in (2, 3):
            self.so, self.sl = self.buf[off + 9], self.buf[off + 10]
            p = off + 12
            self.base = self._u(p, self.so)
            self.root = self._u(p + 3 * self.so, self.so) + self.base
        else:
            raise H5Error("superblock version %d" % ver)

    # ------------------------------------------------------------------ object headers
    def _messages(self, addr: int) -> List[Tuple[int, int, int, int]]:
        """-> [(type, offset of the message data, size, flags)] of the object header at `addr` (continuations followed)."""
        out = []
        if addr < 0 or addr + 16 > len(self.buf):
            raise H5Error("object header address %d lies outside the file (soft / external link or a truncated file)" % addr)
        if self.buf[addr:addr + 4] == b"OHDR":                                  # version 2
            flags = self.buf[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            nb = 1 << (flags & 3)
            size0 = self._u(p, nb)
            p += nb
            blocks = [(p, size0)]
            track = bool(flags & 0x04)
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                while p + 4 <= end:
                    mtype, msize, mflags = self.buf[p], self._u(p + 1, 2), self.buf[p + 3]
                    p += 4 + (2 if track else 0)
                    if mtype == 0x10:
                        caddr, clen = self._addr(p), self._len(p + self.so)
                        if self.buf[caddr:caddr + 4] != b"OCHK":
                            raise H5Error("bad object header continuation block")
                        blocks.append((caddr + 4, clen - 8))                    # signature in front, checksum behind
                    elif mtype != 0:
                        out.append((mtype, p, msize, mflags))
                    p += msize
            return out
        if self.buf[addr] != 1:
            raise H5Error("object header version %d at %d" % (self.buf[addr], addr))
        nmsg, hsize = self._u(addr + 2, 2), self._u(addr + 8, 4)
        blocks = [(addr + 16, hsize)]
        while blocks and len(out) < nmsg + 64:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end:
                mtype, msize, mflags = self._u(p, 2), self._u(p + 2, 2), self.buf[p + 4]
                p += 8
                if mtype == 0x10:
                    blocks.append((self._addr(p), self._len(p + self.so)))
                elif mtype != 0:
                    out.append((mtype, p, msize, mflags))
                p += msize
        return out

    # ------------------------------------------------------------------ groups
    def _links(self, addr: int) -> Dict[str, int]:
        """name -> object header address of the members of the group whose object header is at `addr`."""
        links: Dict[str, int] = {}
        for mtype, p, size, _ in self._messages(addr):
            if mtype == 0x11:                                                   # symbol table: B-tree + local heap
                btree, heap = self._addr(p), self._addr(p + self.so)
                if self.buf[heap:heap + 4] != b"HEAP":
                    raise H5Error("bad local heap")
                data = self._addr(heap + 8 + 2 * self.sl)
                self._walk_group_btree(btree, data, links)
            elif mtype == 0x06:                                                 # link message
                ver, flags = self.buf[p], self.buf[p + 1]
                q = p + 2
                ltype = 0
                if flags & 0x08:
                    ltype = self.buf[q]; q += 1
                if flags & 0x04:
                    q += 8
                if flags & 0x10:
                    q += 1
                nb = 1 << (flags & 3)
                nlen = self._u(q, nb); q += nb
                name = self.buf[q:q + nlen].decode("utf-8"); q += nlen
                if ltype == 0:
                    links[name] = self._addr(q)
            elif mtype == 0x02:                                                 # link info: dense storage when the heap address is set
                q = p + 2 + (8 if self.buf[p + 1] & 1 else 0)
                if self._addr(q) != UNDEF:
                    raise H5Error("dense link storage (a group with many links in a latest-format file)")
        return links

    def _walk_group_btree(self, addr: int, heap_data: int, links: Dict[str, int]):
        if self.buf[addr:addr + 4] != b"TREE" or self.buf[addr + 4] != 0:
            raise H5Error("bad group B-tree node")
        level, used = self.buf[addr + 5], self._u(addr + 6, 2)
        p = addr + 8 + 2 * self.so
        for i in range(used):
            child = self._addr(p + self.sl + i * (self.sl + self.so))
            if level > 0:
                self._walk_group_btree(child, heap_data, links)
                continue
            if self.buf[child:child + 4] != b"SNOD":
                raise H5Error("bad symbol table node")
            n = self._u(child + 6, 2)
            q = child + 8
            for _ in range(n):
                noff = self._u(q, self.so)
                ohdr = self._addr(q + self.so)
                end = self.buf.index(b"\0", heap_data + noff)
                links[self.buf[heap_data + noff:end].decode("utf-8")] = ohdr
                q += 2 * self.so + 24

    def _resolve(self, name: str) -> int:
        addr = self.root
        for part in [p for p in name.split("/") if p]:
            links = self._links(addr)
            if part not in links:
                raise KeyError("%s: no object named %r (have %s)" % (self.path, name, sorted(links)))
            addr = links[part]
        return addr

    def keys(self, group: str = "/") -> List[str]:
        return sorted(self._links(self._resolve(group)))

    def __contains__(self, name: str) -> bool:
        try:
            self._resolve(name)
            return True
        except KeyError:
            return False

    # ------------------------------------------------------------------ datasets
    def _dtype(self, p: int) -> np.dtype:
        cv, bits0 = self.buf[p], self.buf[p + 1]
        cls, size = cv & 0x0F, self._u(p + 4, 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 1 and size in (4, 8):
            return np.dtype(order + "f%d" % size)
        if cls == 0 and size in (1, 2, 4, 8):
            return np.dtype(order + ("i" if bits0 & 0x08 else "u") + "%d" % size)
        raise H5Error("datatype class %d with %d bytes (only IEEE floats and integers are supported)" % (cls, size))

    def _shape(self, p: int) -> Tuple[int, ...]:
        ver, rank = self.buf[p], self.buf[p + 1]
        q = p + (8 if ver == 1 else 4)
        return tuple(self._len(q + i * self.sl) for i in range(rank))

    def _filters(self, p: int) -> List[Tuple[int, List[int]]]:
        ver, n = self.buf[p], self.buf[p + 1]
        q = p + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = self._u(q, 2); q += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = self._u(q, 2); q += 2
            q += 2                                                              # flags
            ncd = self._u(q, 2); q += 2
            if nlen:
                q += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = [self._u(q + 4 * i, 4) for i in range(ncd)]
            q += 4 * ncd
            if ver == 1 and ncd % 2:
                q += 4
            out.append((fid, cd))
        return out

    def _unfilter(self, raw: bytes, filters, mask: int, itemsize: int) -> bytes:
        for i in range(len(filters) - 1, -1, -1):                              # the pipeline is undone in reverse order
            if mask & (1 << i):
                continue
            fid, cd = filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                n = len(raw) // itemsize
                raw = np.frombuffer(raw[:n * itemsize], dtype=np.uint8).reshape(itemsize, n).T.tobytes() + raw[n * itemsize:]
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise H5Error("filter id %d (only deflate, shuffle and fletcher32 are supported)" % fid)
        return raw

    def __getitem__(self, name: str) -> np.ndarray:
        msgs = self._messages(self._resolve(name))
        dtype = shape = layout = None
        filters: List[Tuple[int, List[int]]] = []
        for mtype, p, size, _ in msgs:
            if mtype == 0x03:
                dtype = self._dtype(p)
            elif mtype == 0x01:
                shape = self._shape(p)
            elif mtype == 0x08:
                layout = p
            elif mtype == 0x0B:
                filters = self._filters(p)
        if dtype is None or shape is None or layout is None:
            raise H5Error("%r is not a dataset" % name)
        count = int(np.prod(shape)) if shape else 1
        ver, cls = self.buf[layout], self.buf[layout + 1]
        if ver not in (3, 4):
            raise H5Error("data layout message version %d" % ver)
        if cls == 0:                                                            # compact
            n = self._u(layout + 2, 2)
            return np.frombuffer(self.buf[layout + 4:layout + 4 + n], dtype=dtype, count=count).reshape(shape).astype(dtype.newbyteorder("="))
        if cls == 1:                                                            # contiguous
            addr = self._addr(layout + 2)
            if addr == UNDEF:
                return np.zeros(shape, dtype=dtype.newbyteorder("="))
            return np.frombuffer(self.buf, dtype=dtype, count=count, offset=addr).reshape(shape).astype(dtype.newbyteorder("="))
        if cls != 2:
            raise H5Error("data layout class %d" % cls)
        out = np.zeros(shape, dtype=dtype.newbyteorder("="))
        rank = len(shape)
        if ver == 3:
            nd = self.buf[layout + 2]
            btree = self._addr(layout + 3)
            q = layout + 3 + self.so
            chunk = tuple(self._u(q + 4 * i, 4) for i in range(nd - 1))
            if btree != UNDEF:
                self._walk_chunk_btree(btree, rank, chunk, dtype, filters, out)
            return out
        # version 4 (latest-format files)
        flags, nd = self.buf[layout + 2], self.buf[layout + 3]
        enc = self.buf[layout + 4]
        q = layout + 5
        chunk = tuple(self._u(q + enc * i, enc) for i in range(nd - 1))
        q += enc * nd
        itype = self.buf[q]; q += 1
        nchunks = [-(-s // c) for s, c in zip(shape, chunk)]
        cbytes = int(np.prod(chunk)) * dtype.itemsize
        if itype == 1:                                                          # single chunk
            csize, mask = cbytes, 0
            if flags & 0x02:
                csize = self._len(q); mask = self._u(q + self.sl, 4); q += self.sl + 4
            self._put_chunk(out, (0,) * rank, chunk, self._unfilter(self.buf[self._addr(q):self._addr(q) + csize], filters, mask, dtype.itemsize), dtype)
            return out
        if itype == 2:                                                          # implicit: all chunks back to back, unfiltered
            addr = self._addr(q)
            for i, origin in enumerate(np.ndindex(*nchunks)):
                raw = self.buf[addr + i * cbytes:addr + (i + 1) * cbytes]
                self._put_chunk(out, tuple(o * c for o, c in zip(origin, chunk)), chunk, raw, dtype)
            return out
        if itype == 3:                                                          # fixed array
            page_bits = self.buf[q]; q += 1
            self._fixed_array(self._addr(q), nchunks, chunk, dtype, filters, out, cbytes)
            return out
        raise H5Error("chunk index type %d (extensible array / B-tree v2: datasets with an unlimited dimension)" % itype)

    def _fixed_array(self, hdr: int, nchunks, chunk, dtype, filters, out, cbytes):
        if self.buf[hdr:hdr + 4] != b"FAHD":
            raise H5Error("bad fixed-array header")
        client, esize, page_bits = self.buf[hdr + 5], self.buf[hdr + 6], self.buf[hdr + 7]
        nent = self._len(hdr + 8)
        dblk = self._addr(hdr + 8 + self.sl)
        if self.buf[dblk:dblk + 4] != b"FADB":
            raise H5Error("bad fixed-array data block")
        if nent > (1 << page_bits):
            raise H5Error("paged fixed-array chunk index (more than %d chunks)" % (1 << page_bits))
        p = dblk + 6 + self.so
        for i, origin in enumerate(np.ndindex(*nchunks)):
            e = p + i * esize
            addr = self._addr(e)
            csize, mask = cbytes, 0
            if client == 1:                                                     # filtered chunks: address, size, filter mask
                nb = esize - self.so - 4
                csize, mask = self._u(e + self.so, nb), self._u(e + self.so + nb, 4)
            if addr == UNDEF:
                continue
            raw = self._unfilter(self.buf[addr:addr + csize], filters, mask, dtype.itemsize)
            self._put_chunk(out, tuple(o * c for o, c in zip(origin, chunk)), chunk, raw, dtype)

    def _put_chunk(self, out: np.ndarray, origin, chunk, raw: bytes, dtype: np.dtype):
        block = np.frombuffer(raw, dtype=dtype, count=int(np.prod(chunk))).reshape(chunk)
        sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(origin, chunk, out.shape))
        sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
        out[sel_out] = block[sel_in]

    def _walk_chunk_btree(self, addr: int, rank: int, chunk, dtype, filters, out):
        if self.buf[addr:addr + 4] != b"TREE" or self.buf[addr + 4] != 1:
            raise H5Error("bad chunk B-tree node")
        level, used = self.buf[addr + 5], self._u(addr + 6, 2)
        ksize = 8 + 8 * (rank + 1)
        p = addr + 8 + 2 * self.so
        for i in range(used):
            k = p + i * (ksize + self.so)
            csize, mask = self._u(k, 4), self._u(k + 4, 4)
            origin = tuple(self._u(k + 8 + 8 * d, 8) for d in range(rank))
            child = self._addr(k + ksize)
            if level > 0:
                self._walk_chunk_btree(child, rank, chunk, dtype, filters, out)
            else:
                self._put_chunk(out, origin, chunk, self._unfilter(self.buf[child:child + csize], filters, mask, dtype.itemsize), dtype)


def read(path: str, name: str) -> np.ndarray:
    """`h5py.File(path, 'r')[name][:]`."""
    return File(path)[name]
