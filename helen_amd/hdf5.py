"""Minimal HDF5 access through ctypes on libhdf5 (no h5py in this image).

Only what the polish path's two file formats need:
  - reading MarginPolish image files: group listing, integer / uint8 datasets, short string
    datasets (reference reader: helen/modules/python/models/dataloader_predict.py:38-70);
  - writing prediction files: scalar ints and small integer arrays under nested groups
    (reference writer: helen/modules/python/DataStore.py:99-133).
The library is looked up at run time ($HELEN_LIBHDF5, then the usual locations); importing this
module never fails, opening a file without a usable libhdf5 raises a clear error.
"""
import ctypes
import ctypes.util
import os

import numpy as np

_hid = ctypes.c_int64
_hsize = ctypes.c_uint64
_lib = None

_CANDIDATES = (
    "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so",
    "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/libhdf5.so",
    "/usr/lib64/libhdf5.so", "/usr/local/lib/libhdf5.so",
)

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5S_SCALAR = 0
H5_INDEX_NAME, H5_ITER_INC = 0, 0


class Hdf5Error(IOError):
    pass


class _GInfo(ctypes.Structure):
    _fields_ = [("storage_type", ctypes.c_int), ("nlinks", _hsize), ("max_corder", ctypes.c_int64),
                ("mounted", ctypes.c_int)]


def _load():
    global _lib
    if _lib is not None:
        return _lib
    paths = []
    if os.environ.get("HELEN_LIBHDF5"):
        paths.append(os.environ["HELEN_LIBHDF5"])
    paths.extend(_CANDIDATES)
    found = ctypes.util.find_library("hdf5")
    if found:
        paths.append(found)
    lib = None
    errors = []
    for p in paths:
        try:
            lib = ctypes.CDLL(p)
            break
        except OSError as e:
            errors.append("%s: %s" % (p, e))
    if lib is None:
        raise Hdf5Error("libhdf5 not found (set HELEN_LIBHDF5 to its path). Tried:\n  "
                        + "\n  ".join(errors))
    H, I, P = _hid, ctypes.c_int, ctypes.c_void_p
    sigs = {
        "H5open": (I, []), "H5Eset_auto2": (I, [H, P, P]),
        "H5Fopen": (H, [ctypes.c_char_p, ctypes.c_uint, H]),
        "H5Fcreate": (H, [ctypes.c_char_p, ctypes.c_uint, H, H]),
        "H5Fclose": (I, [H]), "H5Fflush": (I, [H, I]),
        "H5Pset_libver_bounds": (I, [H, I, I]), "H5Pset_fletcher32": (I, [H]),
        "H5get_libversion": (I, [ctypes.POINTER(ctypes.c_uint)] * 3),
        "H5Gopen2": (H, [H, ctypes.c_char_p, H]), "H5Gclose": (I, [H]),
        "H5Gget_info": (I, [H, ctypes.POINTER(_GInfo)]),
        "H5Lget_name_by_idx": (ctypes.c_ssize_t, [H, ctypes.c_char_p, I, I, _hsize, ctypes.c_char_p,
                                                  ctypes.c_size_t, H]),
        "H5Lexists": (I, [H, ctypes.c_char_p, H]),
        "H5Dopen2": (H, [H, ctypes.c_char_p, H]), "H5Dclose": (I, [H]),
        "H5Dget_space": (H, [H]), "H5Dget_type": (H, [H]),
        "H5Dread": (I, [H, H, H, H, H, P]), "H5Dwrite": (I, [H, H, H, H, H, P]),
        "H5Dcreate2": (H, [H, ctypes.c_char_p, H, H, H, H, H]),
        "H5Dvlen_reclaim": (I, [H, H, H, P]),
        "H5Sget_simple_extent_ndims": (I, [H]),
        "H5Sget_simple_extent_dims": (I, [H, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
        "H5Screate_simple": (H, [I, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
        "H5Screate": (H, [I]), "H5Sclose": (I, [H]),
        "H5Tget_class": (I, [H]), "H5Tget_size": (ctypes.c_size_t, [H]), "H5Tget_sign": (I, [H]),
        "H5Tis_variable_str": (I, [H]), "H5Tcopy": (H, [H]), "H5Tclose": (I, [H]),
        "H5Tset_size": (I, [H, ctypes.c_size_t]),
        "H5Pcreate": (H, [H]), "H5Pclose": (I, [H]),
        "H5Pset_create_intermediate_group": (I, [H, ctypes.c_uint]),
        "H5Dget_create_plist": (H, [H]), "H5Pget_layout": (I, [H]), "H5Pget_nfilters": (I, [H]),
        "H5Pget_filter2": (I, [H, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_size_t),
                               P, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint)]),
        "H5Pget_chunk": (I, [H, I, ctypes.POINTER(_hsize)]),
        "H5Pset_chunk": (I, [H, I, ctypes.POINTER(_hsize)]), "H5Pset_deflate": (I, [H, ctypes.c_uint]),
        "H5Pset_shuffle": (I, [H]),
        "H5Zfilter_avail": (I, [I]),
    }
    for name, (res, args) in sigs.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    lib.H5open()
    lib.H5Eset_auto2(0, None, None)   # errors are reported through return codes below
    _lib = lib
    return lib


def _gid(name):
    return _hid.in_dll(_load(), name).value


_NATIVE = {
    np.dtype(np.uint8): "H5T_NATIVE_UINT8_g", np.dtype(np.int8): "H5T_NATIVE_INT8_g",
    np.dtype(np.uint16): "H5T_NATIVE_UINT16_g", np.dtype(np.int16): "H5T_NATIVE_INT16_g",
    np.dtype(np.uint32): "H5T_NATIVE_UINT32_g", np.dtype(np.int32): "H5T_NATIVE_INT32_g",
    np.dtype(np.uint64): "H5T_NATIVE_UINT64_g", np.dtype(np.int64): "H5T_NATIVE_INT64_g",
    np.dtype(np.float32): "H5T_NATIVE_FLOAT_g", np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g",
}
_STD = {
    np.dtype(np.uint8): "H5T_STD_U8LE_g", np.dtype(np.int8): "H5T_STD_I8LE_g",
    np.dtype(np.uint16): "H5T_STD_U16LE_g", np.dtype(np.int16): "H5T_STD_I16LE_g",
    np.dtype(np.uint32): "H5T_STD_U32LE_g", np.dtype(np.int32): "H5T_STD_I32LE_g",
    np.dtype(np.uint64): "H5T_STD_U64LE_g", np.dtype(np.int64): "H5T_STD_I64LE_g",
    np.dtype(np.float32): "H5T_IEEE_F32LE_g", np.dtype(np.float64): "H5T_IEEE_F64LE_g",
}


class File(object):
    """An open HDF5 file.  mode 'r' (read-only) or 'w' (create/truncate)."""

    def __init__(self, path, mode="r", libver=None):
        """libver="latest" (mode 'w' only) writes the newest file format the library knows, as h5py's libver="latest"
        does: superblock 3, version 2 object headers, link-message / dense groups, version 4 layout messages."""
        self._lib = _load()
        self.path = path
        self.mode = mode
        bpath = os.fsencode(path)
        if mode == "r":
            self._fid = self._lib.H5Fopen(bpath, H5F_ACC_RDONLY, 0)
        elif mode == "w":
            fapl = 0
            if libver == "latest":
                fapl = self._lib.H5Pcreate(_gid("H5P_CLS_FILE_ACCESS_ID_g"))
                ver = [ctypes.c_uint() for _ in range(3)]
                self._lib.H5get_libversion(*[ctypes.byref(v) for v in ver])
                # H5F_LIBVER_LATEST: 1 in 1.8, 2 (V110) in 1.10, 3 (V112) in 1.12, 4 (V114) in 1.14
                latest = {8: 1, 10: 2, 12: 3, 14: 4}.get(ver[1].value, 2)
                self._lib.H5Pset_libver_bounds(fapl, latest, latest)
            elif libver is not None:
                raise ValueError("libver must be None or 'latest'")
            self._fid = self._lib.H5Fcreate(bpath, H5F_ACC_TRUNC, 0, fapl)
            if fapl:
                self._lib.H5Pclose(fapl)
            self._lcpl = self._lib.H5Pcreate(_gid("H5P_CLS_LINK_CREATE_ID_g"))
            self._lib.H5Pset_create_intermediate_group(self._lcpl, 1)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self._fid < 0:
            raise Hdf5Error("cannot open HDF5 file '%s' (mode %s)" % (path, mode))

    # ---- lifetime ----
    def close(self):
        if getattr(self, "_fid", -1) >= 0:
            if self.mode == "w":
                self._lib.H5Pclose(self._lcpl)
            self._lib.H5Fclose(self._fid)
            self._fid = -1

    def flush(self):
        if self._fid >= 0:
            self._lib.H5Fflush(self._fid, 1)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- queries ----
    def exists(self, path):
        """True if every component of `path` exists (like `path in h5py_file`)."""
        parts = [p for p in path.split("/") if p]
        cur = ""
        for p in parts:
            cur = cur + "/" + p if cur else p
            if self._lib.H5Lexists(self._fid, cur.encode(), 0) <= 0:
                return False
        return True

    __contains__ = exists

    def keys(self, group="/"):
        """Member names of a group in name order (what h5py's `.keys()` yields)."""
        gid = self._lib.H5Gopen2(self._fid, group.encode(), 0)
        if gid < 0:
            raise Hdf5Error("%s: no group '%s'" % (self.path, group))
        info = _GInfo()
        self._lib.H5Gget_info(gid, ctypes.byref(info))
        names = []
        buf = ctypes.create_string_buffer(1024)
        for i in range(info.nlinks):
            n = self._lib.H5Lget_name_by_idx(gid, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, 1024, 0)
            if n < 0:
                self._lib.H5Gclose(gid)
                raise Hdf5Error("%s: cannot list '%s'" % (self.path, group))
            if n >= 1024:
                big = ctypes.create_string_buffer(n + 1)
                self._lib.H5Lget_name_by_idx(gid, b".", H5_INDEX_NAME, H5_ITER_INC, i, big, n + 1, 0)
                names.append(big.value.decode())
            else:
                names.append(buf.value.decode())
        self._lib.H5Gclose(gid)
        return names

    # ---- reading ----
    def read(self, path, dtype=None):
        """Read a whole dataset.  Numeric -> ndarray (converted to `dtype` by HDF5 if given),
        string -> ndarray of Python str (fixed- and variable-length strings both handled)."""
        L = self._lib
        did = L.H5Dopen2(self._fid, path.encode(), 0)
        if did < 0:
            raise Hdf5Error("%s: no dataset '%s'" % (self.path, path))
        sid = L.H5Dget_space(did)
        tid = L.H5Dget_type(did)
        try:
            nd = L.H5Sget_simple_extent_ndims(sid)
            dims = (_hsize * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sid, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            count = int(np.prod(shape)) if shape else 1
            cls = L.H5Tget_class(tid)
            size = int(L.H5Tget_size(tid))
            if cls == H5T_STRING:
                return self._read_strings(did, tid, shape, count, size)
            if cls not in (H5T_INTEGER, H5T_FLOAT):
                raise Hdf5Error("%s: dataset '%s' has unsupported type class %d" % (self.path, path, cls))
            if dtype is None:
                if cls == H5T_FLOAT:
                    dtype = np.float32 if size == 4 else np.float64
                else:
                    signed = L.H5Tget_sign(tid) != 0
                    dtype = np.dtype("%s%d" % ("i" if signed else "u", size))
            dtype = np.dtype(dtype)
            out = np.empty(shape, dtype=dtype)
            if count and L.H5Dread(did, _gid(_NATIVE[dtype]), 0, 0, 0, out.ctypes.data) < 0:
                raise Hdf5Error("%s: reading '%s' failed" % (self.path, path))
            return out
        finally:
            L.H5Tclose(tid)
            L.H5Sclose(sid)
            L.H5Dclose(did)

    def _read_strings(self, did, tid, shape, count, size):
        L = self._lib
        if L.H5Tis_variable_str(tid) > 0:
            mem = L.H5Tcopy(_gid("H5T_C_S1_g"))
            L.H5Tset_size(mem, ctypes.c_size_t(-1).value)   # H5T_VARIABLE
            ptrs = (ctypes.c_char_p * count)()
            sid = L.H5Dget_space(did)
            if L.H5Dread(did, mem, 0, 0, 0, ptrs) < 0:
                raise Hdf5Error("%s: reading a string dataset failed" % self.path)
            vals = [(p or b"").decode("utf-8", "replace") for p in ptrs]
            L.H5Dvlen_reclaim(mem, sid, 0, ptrs)
            L.H5Sclose(sid)
            L.H5Tclose(mem)
        else:
            mem = L.H5Tcopy(tid)
            buf = ctypes.create_string_buffer(size * max(count, 1))
            if L.H5Dread(did, mem, 0, 0, 0, buf) < 0:
                raise Hdf5Error("%s: reading a string dataset failed" % self.path)
            raw = buf.raw
            vals = [raw[i * size:(i + 1) * size].split(b"\0", 1)[0].decode("utf-8", "replace")
                    for i in range(count)]
            L.H5Tclose(mem)
        return np.array(vals, dtype=object).reshape(shape)

    # ---- writing ----
    def info(self, path):
        """Storage facts of dataset `path`: shape, type class / size / signedness, layout (compact,
        contiguous, chunked), chunk shape and filter ids (1 = deflate, 2 = shuffle, 32000 = lzf, ...)."""
        L = self._lib
        did = L.H5Dopen2(self._fid, path.encode(), 0)
        if did < 0:
            raise Hdf5Error("%s: no dataset '%s'" % (self.path, path))
        sid, tid, pid = L.H5Dget_space(did), L.H5Dget_type(did), L.H5Dget_create_plist(did)
        try:
            nd = L.H5Sget_simple_extent_ndims(sid)
            dims = (_hsize * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sid, dims, None)
            cls = L.H5Tget_class(tid)
            layout = {0: "compact", 1: "contiguous", 2: "chunked"}.get(L.H5Pget_layout(pid), "other")
            chunk = None
            if layout == "chunked" and nd > 0:
                c = (_hsize * nd)()
                L.H5Pget_chunk(pid, nd, c)
                chunk = tuple(int(c[i]) for i in range(nd))
            filters = []
            for i in range(max(0, L.H5Pget_nfilters(pid))):
                flags, nelem, cfg = ctypes.c_uint(0), ctypes.c_size_t(0), ctypes.c_uint(0)
                fid = L.H5Pget_filter2(pid, i, ctypes.byref(flags), ctypes.byref(nelem), None, 0, None,
                                       ctypes.byref(cfg))
                filters.append(int(fid))
            return {"shape": tuple(int(dims[i]) for i in range(nd)),
                    "class": {H5T_INTEGER: "int", H5T_FLOAT: "float", H5T_STRING: "string"}.get(cls, "other"),
                    "size": int(L.H5Tget_size(tid)),
                    "signed": bool(L.H5Tget_sign(tid) == 1) if cls == H5T_INTEGER else None,
                    "variable_string": bool(L.H5Tis_variable_str(tid) > 0) if cls == H5T_STRING else None,
                    "layout": layout, "chunk": chunk, "filters": filters,
                    "filters_available": all(L.H5Zfilter_avail(f) > 0 for f in filters)}
        finally:
            L.H5Pclose(pid)
            L.H5Tclose(tid)
            L.H5Sclose(sid)
            L.H5Dclose(did)

    def write(self, path, value, dtype=None, chunks=None, gzip=None, shuffle=False, string="fixed", fletcher32=False):
        """Create dataset `path` (intermediate groups are created) from a Python int (scalar
        int64 dataset, what `h5py_file[path] = int` makes), a str / bytes (see `string`) or an ndarray
        (a 0-d array makes a scalar dataset).  `chunks` (a shape) makes it chunked, `gzip` (1..9) adds the
        deflate filter (as h5py's compression="gzip" does), `shuffle` the byte-shuffle filter before it.
        string = "fixed" (one-element array of a fixed-length string: numpy 'S'), "vlen" (one-element array of a
        variable-length string: h5py's str), "scalar" / "vlen_scalar" (the same as scalar datasets)."""
        if self.mode != "w":
            raise Hdf5Error("file not opened for writing")
        L = self._lib
        if isinstance(value, (int, np.integer)) and dtype is None:
            arr = np.array(int(value), dtype=np.int64)
        elif isinstance(value, (str, bytes)):
            return self._write_string(path, value, string)
        else:
            arr = np.asarray(value, dtype=dtype)        # (np.ascontiguousarray would turn a 0-d array into a [1] array)
            if not arr.flags.c_contiguous:
                arr = np.ascontiguousarray(arr)
        dt = arr.dtype
        if dt not in _STD:
            raise Hdf5Error("unsupported dtype %s" % dt)
        if arr.ndim == 0:
            sid = L.H5Screate(H5S_SCALAR)
        else:
            dims = (_hsize * arr.ndim)(*arr.shape)
            sid = L.H5Screate_simple(arr.ndim, dims, None)
        dcpl = 0
        if (chunks is not None or gzip or shuffle or fletcher32) and arr.ndim > 0 and arr.size:
            dcpl = L.H5Pcreate(_gid("H5P_CLS_DATASET_CREATE_ID_g"))
            ch = tuple(chunks) if chunks is not None else arr.shape
            L.H5Pset_chunk(dcpl, arr.ndim, (_hsize * arr.ndim)(*[max(1, min(int(c), int(d))) for c, d in zip(ch, arr.shape)]))
            if shuffle:
                L.H5Pset_shuffle(dcpl)
            if gzip:
                L.H5Pset_deflate(dcpl, int(gzip))
            if fletcher32:
                L.H5Pset_fletcher32(dcpl)
        did = L.H5Dcreate2(self._fid, path.encode(), _gid(_STD[dt]), sid, self._lcpl, dcpl, 0)
        if dcpl:
            L.H5Pclose(dcpl)
        if did < 0:
            L.H5Sclose(sid)
            raise Hdf5Error("%s: cannot create dataset '%s' (already exists?)" % (self.path, path))
        rc = L.H5Dwrite(did, _gid(_NATIVE[dt]), 0, 0, 0, arr.ctypes.data) if arr.size else 0
        L.H5Dclose(did)
        L.H5Sclose(sid)
        if rc < 0:
            raise Hdf5Error("%s: writing '%s' failed" % (self.path, path))

    def _write_string(self, path, value, kind="fixed"):
        """A string dataset of one element: fixed-length (how numpy 'S' arrays and the synthetic image files
        store `contig`) or variable-length (how h5py stores Python str), as a [1] array or a scalar."""
        L = self._lib
        data = value.encode() if isinstance(value, str) else value
        vlen = kind.startswith("vlen")
        size = max(len(data), 1)
        tid = L.H5Tcopy(_gid("H5T_C_S1_g"))
        L.H5Tset_size(tid, ctypes.c_size_t(-1).value if vlen else size)
        if kind in ("scalar", "vlen_scalar"):
            sid = L.H5Screate(H5S_SCALAR)
        else:
            dims = (_hsize * 1)(1)
            sid = L.H5Screate_simple(1, dims, None)
        did = L.H5Dcreate2(self._fid, path.encode(), tid, sid, self._lcpl, 0, 0)
        if did < 0:
            raise Hdf5Error("%s: cannot create dataset '%s'" % (self.path, path))
        if vlen:
            keep = ctypes.create_string_buffer(data)
            buf = (ctypes.c_char_p * 1)(ctypes.cast(keep, ctypes.c_char_p))
        else:
            buf = ctypes.create_string_buffer(data, size)
        rc = L.H5Dwrite(did, tid, 0, 0, 0, buf)
        L.H5Dclose(did)
        L.H5Sclose(sid)
        L.H5Tclose(tid)
        if rc < 0:
            raise Hdf5Error("%s: writing '%s' failed" % (self.path, path))


def available():
    """True if a usable libhdf5 can be loaded."""
    try:
        _load()
        return True
    except Hdf5Error:
        return False
