#!/usr/bin/env python3
"""Minimal reader for real, dense, double-precision variables of a MATLAB v7.3 MAT-file (= HDF5), without h5py.

Why: the reference ships the estimator's linear model `model_approx.mat` (variables A_s, b_s; loaded at
/root/reference/README.md:294 and used at README.md:478 `ad_est = lsqminnorm(A_s'*A_s, A_s'*(Y_M - b_s))`) as the one piece
of reference-held NUMERICAL DATA on a SURVEY §8 row.  This script turns it into the fixture
`tests/golden/model_approx_As_bs.npz` (data, not source), so that the estimator's linear half is pinned to the reference's
own numbers on the GPU box, where /root/reference does not exist.

Only what such a file needs is implemented (HDF5 File Format Specification, version 0 structures):
  * user block of 512 bytes (the MATLAB text header), superblock version 0, 8-byte offsets and lengths
  * old-style groups: symbol-table entry -> B-tree (node type 0) -> symbol-table nodes (SNOD) -> names in a local heap
  * version-1 object headers with continuation blocks; messages: dataspace (1), datatype (3), filter pipeline (11),
    data layout (8) version 3, contiguous or chunked (B-tree node type 1)
  * filters: deflate (zlib) and shuffle
  * datatype: little-endian IEEE floating point of 8 bytes
MATLAB stores an r x c array column-major, i.e. as an HDF5 dataset of shape (c, r): `read_mat73` returns the array in
MATLAB's orientation (transposed back).

  python tests/golden/read_mat73.py [/root/reference/model_approx.mat [tests/golden/model_approx_As_bs.npz]]
"""
import os
import struct
import sys
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class _H5:
    def __init__(self, buf):
        self.b = buf
        sig = b"\x89HDF\r\n\x1a\n"
        self.base = None
        off = 0
        while off < len(buf):                               # the superblock sits at 0, 512, 1024, ... (user block in front)
            if buf[off:off + 8] == sig:
                self.base = off
                break
            off = 512 if off == 0 else off * 2
        if self.base is None:
            raise ValueError("no HDF5 superblock found")
        sb = self.base
        ver = buf[sb + 8]
        if ver != 0:
            raise ValueError("superblock version %d not supported" % ver)
        self.so, self.sl = buf[sb + 13], buf[sb + 14]
        if (self.so, self.sl) != (8, 8):
            raise ValueError("only 8-byte offsets / lengths are supported")
        # 8 sig, 8 version bytes, 2+2 group K, 4 flags, then base, free space, EOF, driver info addresses
        p = sb + 24
        self.base_addr, _, self.eof, _ = struct.unpack_from("<4Q", buf, p)
        p += 32
        # root group symbol-table entry: name offset, header address, cache type, reserved, scratch (B-tree, heap)
        _, self.root_hdr, cache, _ = struct.unpack_from("<QQII", buf, p)
        self.root_btree, self.root_heap = struct.unpack_from("<QQ", buf, p + 24) if cache == 1 else (None, None)
        self.user = 0       # addresses in the file are relative to the base address; MATLAB writes base address = 512?

    def a(self, addr):
        """file offset of an HDF5 address (relative to the superblock's base address)"""
        return addr + self.base_addr

    # ---------------------------------------------------------------- object headers
    def messages(self, hdr_addr):
        b = self.b
        p = self.a(hdr_addr)
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", b, p)
        if ver != 1:
            raise ValueError("object header version %d not supported" % ver)
        blocks = [(p + 16, hsize)]                           # 12 bytes of prefix, padded to 16
        out = []
        while blocks and len(out) < nmsg:
            q, left = blocks.pop(0)
            end = q + left
            while q + 8 <= end and len(out) < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, q)
                body = b[q + 8:q + 8 + msize]
                q += 8 + msize
                if mtype == 0x10:                            # continuation: offset, length
                    co, cl = struct.unpack_from("<QQ", body, 0)
                    blocks.append((self.a(co), cl))
                out.append((mtype, body))
        return out

    # ---------------------------------------------------------------- groups
    def _heap_data(self, heap_addr):
        p = self.a(heap_addr)
        if self.b[p:p + 4] != b"HEAP":
            raise ValueError("local heap signature missing")
        _, _, data_addr = struct.unpack_from("<QQQ", self.b, p + 8)
        return self.a(data_addr)

    def _walk_group_btree(self, addr, heap_data, out):
        b = self.b
        p = self.a(addr)
        if b[p:p + 4] == b"SNOD":
            nsym = struct.unpack_from("<H", b, p + 6)[0]
            q = p + 8
            for _ in range(nsym):
                name_off, hdr, cache = struct.unpack_from("<QQI", b, q)
                e = b.index(b"\0", heap_data + name_off)
                out[b[heap_data + name_off:e].decode()] = hdr
                q += 40
            return
        if b[p:p + 4] != b"TREE":
            raise ValueError("group B-tree signature missing")
        ntype, level, used = struct.unpack_from("<BBH", b, p + 4)
        if ntype != 0:
            raise ValueError("not a group B-tree")
        q = p + 24                                            # keys (8 bytes) and children (8 bytes) alternate
        for i in range(used):
            child = struct.unpack_from("<Q", b, q + 8)[0]
            self._walk_group_btree(child, heap_data, out)
            q += 16

    def root_entries(self):
        bt, hp = self.root_btree, self.root_heap
        if bt is None:                                        # not cached: take the symbol-table message of the root header
            for mtype, body in self.messages(self.root_hdr):
                if mtype == 0x11:
                    bt, hp = struct.unpack_from("<QQ", body, 0)
        out = {}
        self._walk_group_btree(bt, self._heap_data(hp), out)
        return out

    # ---------------------------------------------------------------- datasets
    def _walk_chunk_btree(self, addr, ndim, out):
        b = self.b
        p = self.a(addr)
        if b[p:p + 4] != b"TREE":
            raise ValueError("chunk B-tree signature missing")
        ntype, level, used = struct.unpack_from("<BBH", b, p + 4)
        if ntype != 1:
            raise ValueError("not a chunk B-tree")
        q = p + 24
        ksz = 8 + 8 * ndim
        for i in range(used):
            csize, fmask = struct.unpack_from("<II", b, q)
            offs = struct.unpack_from("<%dQ" % ndim, b, q + 8)
            child = struct.unpack_from("<Q", b, q + ksz)[0]
            if level == 0:
                out.append((offs[:-1], csize, fmask, child))
            else:
                self._walk_chunk_btree(child, ndim, out)
            q += ksz + 8

    def dataset(self, hdr_addr):
        shape = dtype = layout = None
        filters = []
        for mtype, body in self.messages(hdr_addr):
            if mtype == 0x01:
                ver, rank, flags = struct.unpack_from("<BBB", body, 0)
                if ver == 1:
                    shape = struct.unpack_from("<%dQ" % rank, body, 8)
                elif ver == 2:
                    shape = struct.unpack_from("<%dQ" % rank, body, 4)
                else:
                    raise ValueError("dataspace version %d" % ver)
            elif mtype == 0x03:
                cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", body, 0)
                cls = cv & 0x0F
                if cls != 1 or size != 8 or (b0 & 1):
                    raise ValueError("only little-endian 8-byte floating point is supported (class %d, size %d)" % (cls, size))
                dtype = np.dtype("<f8")
            elif mtype == 0x0B:
                ver, nf = struct.unpack_from("<BB", body, 0)
                if ver != 1:
                    raise ValueError("filter pipeline version %d" % ver)
                q = 8
                for _ in range(nf):
                    fid, nlen, fflags, ncd = struct.unpack_from("<HHHH", body, q)
                    q += 8 + ((nlen + 7) // 8) * 8
                    cd = struct.unpack_from("<%dI" % ncd, body, q)
                    q += 4 * ncd + (4 if ncd % 2 else 0)
                    filters.append((fid, cd))
            elif mtype == 0x08:
                ver, lclass = struct.unpack_from("<BB", body, 0)
                if ver != 3:
                    raise ValueError("data layout version %d" % ver)
                if lclass == 1:
                    addr, size = struct.unpack_from("<QQ", body, 2)
                    layout = ("contiguous", addr, size)
                elif lclass == 2:
                    nd = body[2]
                    addr = struct.unpack_from("<Q", body, 3)[0]
                    cdims = struct.unpack_from("<%dI" % nd, body, 11)
                    layout = ("chunked", addr, cdims)
                else:
                    raise ValueError("data layout class %d" % lclass)
        if shape is None or dtype is None or layout is None:
            raise ValueError("not a simple dataset")
        out = np.zeros(shape, dtype=dtype)
        if layout[0] == "contiguous":
            p = self.a(layout[1])
            out[...] = np.frombuffer(self.b, dtype=dtype, count=int(np.prod(shape)), offset=p).reshape(shape)
            return out
        _, bt, cdims = layout
        cshape = cdims[:-1]                                    # the last entry is the element size
        chunks = []
        self._walk_chunk_btree(bt, len(cdims), chunks)
        for offs, csize, fmask, addr in chunks:
            raw = self.b[self.a(addr):self.a(addr) + csize]
            for idx in reversed(range(len(filters))):          # undo the pipeline back to front
                if fmask & (1 << idx):
                    continue
                fid, cd = filters[idx]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                                 # shuffle: bytes of the elements de-interleaved
                    es = cd[0] if cd else dtype.itemsize
                    raw = np.frombuffer(raw, dtype=np.uint8).reshape(es, -1).T.tobytes()
                else:
                    raise ValueError("filter %d not supported" % fid)
            blk = np.frombuffer(raw, dtype=dtype, count=int(np.prod(cshape))).reshape(cshape)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, shape))
            out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out


def read_mat73(path):
    """{name: ndarray in MATLAB's orientation} for every real double variable at the root of a v7.3 MAT-file."""
    with open(path, "rb") as f:
        buf = f.read()
    if not buf.startswith(b"MATLAB 7.3 MAT-file"):
        raise ValueError("not a MATLAB v7.3 MAT-file")
    h = _H5(buf)
    out = {}
    for name, hdr in h.root_entries().items():
        if name.startswith("#"):                               # #refs#, #subsystem#: not plain variables
            continue
        out[name] = np.ascontiguousarray(h.dataset(hdr).T)     # HDF5 (c, r) row-major == MATLAB r x c column-major
    return out


def main(argv):
    src = argv[1] if len(argv) > 1 else "/root/reference/model_approx.mat"
    here = os.path.dirname(os.path.abspath(__file__))
    dst = argv[2] if len(argv) > 2 else os.path.join(here, "model_approx_As_bs.npz")
    v = read_mat73(src)
    A_s, b_s = v["A_s"], v["b_s"]
    print("A_s", A_s.shape, "b_s", b_s.shape)
    G = A_s.T @ A_s
    ev = np.linalg.eigvalsh(G)
    print("rank(A_s) =", np.linalg.matrix_rank(A_s), " cond(A_s'A_s) = %.3e" % (ev[-1] / ev[0]),
          " eig range [%.3e, %.3e]" % (ev[0], ev[-1]))
    np.savez_compressed(dst, A_s=A_s, b_s=b_s)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main(sys.argv)
