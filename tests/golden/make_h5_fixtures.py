#!/usr/bin/env python3
"""Write the small HDF5 fixtures under tests/golden/h5/ with the REAL HDF5 library (libhdf5 through ctypes -- the library h5py wraps;
h5py itself is not installed in the build image).  The files mimic how the reference's datasets are stored
(`h5py.File(...).create_dataset('poisson_<np>', data=...)`, Generation/H5DataLoader.py:14-17) in the layouts h5py can produce:
contiguous, chunked, chunked + gzip (+ shuffle), float32 / float64, old-style (superblock 0) and latest-format (superblock 3) files.
tests/test_h5_reader.py checks spgan.h5lite against the arrays stored next to them (h5/expected.npz).  Runs only where a libhdf5.so
exists (here: /opt/conda/lib)."""
import ctypes as C
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "h5")


def lib():
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            try:
                return C.CDLL(p)
            except OSError:
                pass
    raise SystemExit("no libhdf5 found")


def main():
    h = lib()
    hid = C.c_int64
    for f in ("H5Fcreate", "H5Screate_simple", "H5Dcreate2", "H5Pcreate", "H5Gcreate2"):
        getattr(h, f).restype = hid
    h.H5open()
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    h.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
    print("libhdf5 %d.%d.%d" % (maj.value, mnr.value, rel.value))
    g = lambda name: hid.in_dll(h, name).value
    F32, F64, I32 = g("H5T_NATIVE_FLOAT_g"), g("H5T_NATIVE_DOUBLE_g"), g("H5T_NATIVE_INT_g")
    F32BE = g("H5T_IEEE_F32BE_g")
    DCPL, FAPL = g("H5P_CLS_DATASET_CREATE_ID_g"), g("H5P_CLS_FILE_ACCESS_ID_g")
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)
    expected = {}

    def write(fname, datasets, latest=False):
        fapl = hid(0)
        if latest:
            fapl = hid(h.H5Pcreate(hid(FAPL)))
            assert h.H5Pset_libver_bounds(fapl, 2, 2) >= 0                    # H5F_LIBVER_LATEST in 1.10 = 2 (V110)
        fid = hid(h.H5Fcreate(os.path.join(OUT, fname).encode(), 2, hid(0), fapl))   # H5F_ACC_TRUNC
        assert fid.value >= 0
        for name, arr, opts in datasets:
            dims = (C.c_uint64 * arr.ndim)(*arr.shape)
            sid = hid(h.H5Screate_simple(arr.ndim, dims, None))
            dcpl = hid(h.H5Pcreate(hid(DCPL)))
            if "chunks" in opts:
                ch = (C.c_uint64 * arr.ndim)(*opts["chunks"])
                assert h.H5Pset_chunk(dcpl, arr.ndim, ch) >= 0
                if opts.get("shuffle"):
                    assert h.H5Pset_shuffle(dcpl) >= 0
                if "gzip" in opts:
                    assert h.H5Pset_deflate(dcpl, opts["gzip"]) >= 0
                if opts.get("fletcher32"):
                    assert h.H5Pset_fletcher32(dcpl) >= 0
            mem = {np.dtype("float32"): F32, np.dtype("float64"): F64, np.dtype("int32"): I32}[arr.dtype]
            ftype = opts.get("filetype", mem)
            parent = fid
            if "/" in name:
                grp, name = name.split("/")
                parent = hid(h.H5Gcreate2(fid, grp.encode(), hid(0), hid(0), hid(0)))
            did = hid(h.H5Dcreate2(parent, name.encode(), hid(ftype), sid, hid(0), dcpl, hid(0)))
            assert did.value >= 0, name
            a = np.ascontiguousarray(arr)
            assert h.H5Dwrite(did, hid(mem), hid(0), hid(0), hid(0), a.ctypes.data_as(C.c_void_p)) >= 0
            h.H5Dclose(did); h.H5Pclose(dcpl); h.H5Sclose(sid)
            if parent is not fid:
                h.H5Gclose(parent)
        h.H5Fclose(fid)
        for name, arr, opts in datasets:
            expected["%s|%s" % (fname, name)] = arr

    pts = lambda s, p, dt=np.float32: (rng.standard_normal((s, p, 3)) * 0.3).astype(dt)
    a = pts(6, 64)
    write("chair_contiguous.h5", [("poisson_64", a, {}), ("poisson_32", a[:, :32].copy(), {})])
    write("chair_f64.h5", [("poisson_64", pts(4, 64, np.float64), {})])
    write("chair_chunked.h5", [("poisson_64", pts(9, 64), {"chunks": (4, 64, 3)})])                       # ragged last chunk
    write("chair_gzip.h5", [("poisson_64", pts(10, 64), {"chunks": (3, 32, 3), "gzip": 4, "shuffle": True})])
    write("chair_gzip_fletcher.h5", [("poisson_64", pts(5, 64), {"chunks": (5, 64, 3), "gzip": 9, "fletcher32": True})])
    write("chair_bigendian.h5", [("poisson_64", pts(3, 64), {"filetype": F32BE})])
    write("many_datasets.h5", [("poisson_%d" % n, pts(2, n), {}) for n in (8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 128)]
          + [("labels", np.arange(12, dtype=np.int32), {}), ("grp/poisson_64", pts(2, 64), {})])
    write("many_chunks.h5", [("poisson_8", pts(150, 8), {"chunks": (1, 8, 3), "gzip": 1})])                # > 64 chunks: two B-tree levels
    write("latest_format.h5", [("poisson_64", pts(7, 64), {}), ("poisson_32", pts(7, 32), {"chunks": (7, 32, 3), "gzip": 1})], latest=True)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
    for f in sorted(os.listdir(OUT)):
        print("%-28s %7d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
