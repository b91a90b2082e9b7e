"""HDF5 modality files <-> HBM, over the HDF5 C library (ctypes; h5py is not needed).

File layout = what the reference's converter writes and its HDF5 importer reads (reference csv_to_h5.py:13-21,
:107-113; h5_dataloader.py:88-116):

    /matrix         (n_samples, n_features) float32, chunked (1, n_features)
    /sample_ids     (n_samples,)  fixed-length byte strings
    /feature_names  (n_features,) fixed-length byte strings

``read_matrix_to_device`` streams ``/matrix`` by row blocks straight into pinned host buffers (``H5Dread`` with a
hyperslab selection writes into the pinned memory, so there is no pageable intermediate) and DMAs each block into
its place in the HBM-resident matrix while the next block is being read.  The result, samples as rows, is exactly
what ``DeviceImporter.import_matrices`` takes -- the reference's transpose to features-as-rows and back
(h5_dataloader.py:104-108, data.py:549) never happens.

The reference reads these files through h5py, which this image lacks, so the container format is exercised against
the HDF5 library itself (HDF5 1.10.x here): files written by ``write_modality_h5`` (same datasets, dtypes and chunking
as csv_to_h5.py) are checked with the library's own ``h5dump`` and read back (tests/test_h5io.py).
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0x0000, 0x0002
H5P_DEFAULT, H5S_ALL = 0, 0
H5S_SELECT_SET = 0
H5T_FLOAT, H5T_STRING = 1, 3          # H5T_class_t
H5D_CHUNKED = 2                        # H5D_layout_t


class H5Error(RuntimeError):
    pass


_lib = None


def _find_lib() -> str:
    cands = []
    if os.environ.get("FX_HDF5_LIB"):
        cands.append(os.environ["FX_HDF5_LIB"])
    found = ctypes.util.find_library("hdf5")
    if found:
        cands.append(found)
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib64/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*"):
        cands += sorted(glob.glob(pat))
    for c in cands:
        try:
            C.CDLL(c)
            return c
        except OSError:
            continue
    raise H5Error("HDF5 C library not found (set FX_HDF5_LIB to libhdf5.so); .h5 modality files cannot be read")


def lib():
    """The loaded libhdf5 with argument types set (hid_t is 64-bit since HDF5 1.10)."""
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(_find_lib())
    sig = {
        "H5open": (herr_t, []),
        "H5get_libversion": (herr_t, [C.POINTER(C.c_uint)] * 3),
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dget_create_plist": (hid_t, [hid_t]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t),
                                         C.POINTER(hsize_t)]),
        "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
        "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]),
        "H5Tclose": (herr_t, [hid_t]),
        "H5Tis_variable_str": (C.c_int, [hid_t]),
        "H5Pcreate": (hid_t, [hid_t]),
        "H5Pset_chunk": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pget_layout": (C.c_int, [hid_t]),
        "H5Pget_chunk": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pclose": (herr_t, [hid_t]),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    # releasing variable-length strings: H5Treclaim (HDF5 >= 1.12) or the deprecated H5Dvlen_reclaim, which builds made with
    # --disable-deprecated-symbols lack.  Bound if present; required only inside the variable-length path.
    L._fx_reclaim = None
    for name in ("H5Treclaim", "H5Dvlen_reclaim"):
        try:
            fn = getattr(L, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = herr_t, [hid_t, hid_t, hid_t, C.c_void_p]
        L._fx_reclaim = fn
        break
    if L.H5open() < 0:
        raise H5Error("H5open failed")
    L.H5Eset_auto2(0, None, None)          # errors are reported through return codes -> H5Error, not stderr dumps
    _lib = L
    return L


def _g(name: str) -> int:
    """Library globals behind the H5T_* / H5P_* macros (valid after H5open)."""
    return hid_t.in_dll(lib(), name).value


def _chk(v: int, what: str) -> int:
    if v < 0:
        raise H5Error(f"HDF5: {what} failed")
    return v


def _dims(space: int) -> Tuple[int, ...]:
    L = lib()
    nd = _chk(L.H5Sget_simple_extent_ndims(space), "H5Sget_simple_extent_ndims")
    d = (hsize_t * max(nd, 1))()
    _chk(L.H5Sget_simple_extent_dims(space, d, None), "H5Sget_simple_extent_dims")
    return tuple(int(d[i]) for i in range(nd))


class H5File:
    """Minimal context manager over a file id."""

    def __init__(self, path: str, mode: str = "r"):
        L = lib()
        p = os.fsencode(path)
        if mode == "r":
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            self.id = L.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            self.id = L.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("mode is 'r' or 'w'")
        if self.id < 0:
            raise H5Error(f"cannot open {path!r} as HDF5 (mode {mode})")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        lib().H5Fclose(self.id)
        return False

    # ---- reading -------------------------------------------------------------------------------------------------
    def matrix_info(self, name: str = "matrix"):
        """(shape, numpy dtype, chunk shape or None) of a 2-D float dataset."""
        L = lib()
        d = _chk(L.H5Dopen2(self.id, name.encode(), H5P_DEFAULT), f"open dataset /{name}")
        try:
            sp = _chk(L.H5Dget_space(d), "H5Dget_space")
            shape = _dims(sp)
            L.H5Sclose(sp)
            t = _chk(L.H5Dget_type(d), "H5Dget_type")
            cls, size = L.H5Tget_class(t), L.H5Tget_size(t)
            L.H5Tclose(t)
            if cls != H5T_FLOAT or size not in (4, 8) or len(shape) != 2:
                raise H5Error(f"/{name}: expected a 2-D float32/float64 dataset, got class {cls} size {size} shape {shape}")
            pl = _chk(L.H5Dget_create_plist(d), "H5Dget_create_plist")
            chunk = None
            if L.H5Pget_layout(pl) == H5D_CHUNKED:
                c = (hsize_t * 2)()
                L.H5Pget_chunk(pl, 2, c)
                chunk = (int(c[0]), int(c[1]))
            L.H5Pclose(pl)
            return shape, (np.float32 if size == 4 else np.float64), chunk
        finally:
            L.H5Dclose(d)

    def read_rows_into(self, buf_ptr: int, r0: int, r1: int, n_cols: int, np_dtype, name: str = "matrix"):
        """Rows [r0, r1) of the 2-D dataset -> contiguous memory at ``buf_ptr`` (any host memory, e.g. pinned)."""
        L = lib()
        d = _chk(L.H5Dopen2(self.id, name.encode(), H5P_DEFAULT), f"open dataset /{name}")
        try:
            fsp = _chk(L.H5Dget_space(d), "H5Dget_space")
            start = (hsize_t * 2)(r0, 0)
            count = (hsize_t * 2)(r1 - r0, n_cols)
            _chk(L.H5Sselect_hyperslab(fsp, H5S_SELECT_SET, start, None, count, None), "H5Sselect_hyperslab")
            msp = _chk(L.H5Screate_simple(2, count, None), "H5Screate_simple")
            mt = _g("H5T_NATIVE_FLOAT_g") if np_dtype == np.float32 else _g("H5T_NATIVE_DOUBLE_g")
            rc = L.H5Dread(d, mt, msp, fsp, H5P_DEFAULT, C.c_void_p(buf_ptr))
            L.H5Sclose(msp)
            L.H5Sclose(fsp)
            _chk(rc, f"H5Dread /{name} rows {r0}:{r1}")
        finally:
            L.H5Dclose(d)

    def read_strings(self, name: str) -> List[str]:
        """A 1-D string dataset -> list of str (reference h5_dataloader.py:101-102).  The converter writes fixed-length
        byte strings (csv_to_h5.py:111-112, dtype 'S'); h5py writes ``str`` data as VARIABLE-length strings, which the
        library hands back as an array of char pointers -- both are handled."""
        L = lib()
        d = _chk(L.H5Dopen2(self.id, name.encode(), H5P_DEFAULT), f"open dataset /{name}")
        try:
            sp = _chk(L.H5Dget_space(d), "H5Dget_space")
            (n,) = _dims(sp)
            L.H5Sclose(sp)
            t = _chk(L.H5Dget_type(d), "H5Dget_type")
            if L.H5Tget_class(t) != H5T_STRING:
                L.H5Tclose(t)
                raise H5Error(f"/{name}: expected a string dataset")
            if L.H5Tis_variable_str(t) > 0:
                L.H5Tclose(t)
                mt = _chk(L.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
                try:
                    _chk(L.H5Tset_size(mt, C.c_size_t(-1).value), "H5Tset_size(H5T_VARIABLE)")
                    ptrs = (C.c_char_p * max(n, 1))()
                    sp = _chk(L.H5Dget_space(d), "H5Dget_space")
                    try:
                        _chk(L.H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, C.cast(ptrs, C.c_void_p)), f"H5Dread /{name}")
                        out = [(ptrs[i] or b"").decode() for i in range(n)]
                        if L._fx_reclaim is None:
                            raise H5Error("this libhdf5 exports neither H5Treclaim nor H5Dvlen_reclaim: variable-length strings "
                                          "cannot be released")
                        _chk(L._fx_reclaim(mt, sp, H5P_DEFAULT, C.cast(ptrs, C.c_void_p)), "vlen reclaim")   # the library allocated the strings
                    finally:
                        L.H5Sclose(sp)
                    return out
                finally:
                    L.H5Tclose(mt)
            size = L.H5Tget_size(t)
            buf = C.create_string_buffer(max(n * size, 1))
            rc = L.H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, C.cast(buf, C.c_void_p))
            L.H5Tclose(t)
            _chk(rc, f"H5Dread /{name}")
            raw = buf.raw
            return [raw[i * size:(i + 1) * size].rstrip(b"\x00").decode() for i in range(n)]
        finally:
            L.H5Dclose(d)

    # ---- writing (the converter's layout, reference csv_to_h5.py:107-113) --------------------------------------------
    def write_matrix(self, arr: np.ndarray, name: str = "matrix", chunks: Optional[Tuple[int, int]] = None):
        L = lib()
        arr = np.ascontiguousarray(arr)
        if arr.dtype not in (np.float32, np.float64) or arr.ndim != 2:
            raise H5Error("write_matrix: 2-D float32/float64 array expected")
        dims = (hsize_t * 2)(*arr.shape)
        sp = _chk(L.H5Screate_simple(2, dims, None), "H5Screate_simple")
        pl = H5P_DEFAULT
        if chunks is not None:
            pl = _chk(L.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g")), "H5Pcreate")
            _chk(L.H5Pset_chunk(pl, 2, (hsize_t * 2)(*chunks)), "H5Pset_chunk")
        t = _g("H5T_NATIVE_FLOAT_g") if arr.dtype == np.float32 else _g("H5T_NATIVE_DOUBLE_g")
        d = _chk(L.H5Dcreate2(self.id, name.encode(), t, sp, H5P_DEFAULT, pl, H5P_DEFAULT), f"create /{name}")
        rc = L.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p))
        L.H5Dclose(d)
        if pl != H5P_DEFAULT:
            L.H5Pclose(pl)
        L.H5Sclose(sp)
        _chk(rc, f"H5Dwrite /{name}")

    def write_strings(self, name: str, values: Sequence[str], variable: bool = False):
        """Fixed-length byte strings (the converter's dtype 'S', default) or, with ``variable``, the variable-length
        strings h5py writes for ``str`` data."""
        L = lib()
        enc = [str(v).encode() for v in values]
        if variable:
            t = _chk(L.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
            _chk(L.H5Tset_size(t, C.c_size_t(-1).value), "H5Tset_size(H5T_VARIABLE)")
            ptrs = (C.c_char_p * max(len(enc), 1))(*enc)
            dims = (hsize_t * 1)(len(enc))
            sp = _chk(L.H5Screate_simple(1, dims, None), "H5Screate_simple")
            d = _chk(L.H5Dcreate2(self.id, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create /{name}")
            rc = L.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, C.cast(ptrs, C.c_void_p))
            L.H5Dclose(d)
            L.H5Sclose(sp)
            L.H5Tclose(t)
            _chk(rc, f"H5Dwrite /{name}")
            return
        size = max([len(e) for e in enc] + [1])
        arr = np.array(enc, dtype=f"S{size}")                      # numpy 'S' = fixed-length, NUL padded
        t = _chk(L.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
        _chk(L.H5Tset_size(t, size), "H5Tset_size")
        dims = (hsize_t * 1)(len(enc))
        sp = _chk(L.H5Screate_simple(1, dims, None), "H5Screate_simple")
        d = _chk(L.H5Dcreate2(self.id, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create /{name}")
        rc = L.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p))
        L.H5Dclose(d)
        L.H5Sclose(sp)
        L.H5Tclose(t)
        _chk(rc, f"H5Dwrite /{name}")


def write_modality_h5(path: str, matrix: np.ndarray, sample_ids: Sequence[str], feature_names: Sequence[str]):
    """Write one modality in the converter's layout (reference csv_to_h5.py:107-113): ``matrix`` is
    ``[n_samples, n_features]``, stored float32 with one chunk per sample row."""
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    if m.shape != (len(sample_ids), len(feature_names)):
        raise H5Error(f"matrix {m.shape} does not match {len(sample_ids)} samples x {len(feature_names)} features")
    with H5File(path, "w") as f:
        f.write_matrix(m, "matrix", chunks=(1, m.shape[1]))
        f.write_strings("sample_ids", sample_ids)
        f.write_strings("feature_names", feature_names)


def read_modality_h5(path: str):
    """Host-side read: (matrix [n_samples, n_features] in its stored dtype, sample_ids, feature_names)."""
    with H5File(path) as f:
        (n, F), dt, _ = f.matrix_info()
        out = np.empty((n, F), dtype=dt)
        if n and F:
            f.read_rows_into(out.ctypes.data, 0, n, F, dt)
        return out, f.read_strings("sample_ids"), f.read_strings("feature_names")


_PINS: list = []


def _pinned_pair(nbytes: int):
    """Two page-locked staging buffers, kept for the life of the process (allocating 2 x 64 MB of pinned memory costs
    more than reading a 164 MB matrix)."""
    import torch
    global _PINS
    if not _PINS or _PINS[0].numel() < nbytes:
        _PINS = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
    return _PINS


def read_matrix_to_device(path: str, device="cuda:0", block_bytes: int = 64 << 20):
    """``/matrix`` -> HBM through two pinned staging buffers: the HDF5 library reads row block i+1 from the file into
    one buffer while the DMA engine copies block i out of the other.  Returns (device tensor [n_samples, n_features]
    in the stored dtype, sample_ids, feature_names)."""
    import torch
    dev = torch.device(device)
    if dev.type != "cuda":
        raise H5Error("read_matrix_to_device targets GPU memory")
    with H5File(path) as f:
        (n, F), dt, _ = f.matrix_info()
        tdt = torch.float32 if dt == np.float32 else torch.float64
        dst = torch.empty((n, F), dtype=tdt, device=dev)
        samples, feats = f.read_strings("sample_ids"), f.read_strings("feature_names")
        if n == 0 or F == 0:
            return dst, samples, feats
        row_bytes = F * dst.element_size()
        rows_per = max(1, min(n, block_bytes // row_bytes))
        pins = _pinned_pair(rows_per * row_bytes)
        stream = torch.cuda.Stream(device=dev)
        # dst may be a block the caching allocator just recycled: work queued on the current stream can still touch it
        stream.wait_stream(torch.cuda.current_stream(dev))
        done = [None, None]
        for k, r0 in enumerate(range(0, n, rows_per)):
            r1 = min(n, r0 + rows_per)
            b = k & 1
            if done[b] is not None:
                done[b].synchronize()                              # the DMA that last read this buffer has finished
            f.read_rows_into(pins[b].data_ptr(), r0, r1, F, dt)     # file -> pinned memory, no pageable copy
            stage = pins[b][: (r1 - r0) * row_bytes].view(tdt).view(r1 - r0, F)
            with torch.cuda.stream(stream):
                dst[r0:r1].copy_(stage, non_blocking=True)
                done[b] = torch.cuda.Event()
                done[b].record(stream)
        for ev in done:
            if ev is not None:
                ev.synchronize()
        torch.cuda.current_stream(dev).wait_stream(stream)
        return dst, samples, feats
