"""Minimal HDF5 writer / reader over the HDF5 C library through ctypes -- used for the result
files when neither PyTables (what the reference's drivers call) nor h5py is installed but
libhdf5 itself is (as in the ROCm image: /opt/conda/lib/libhdf5.so).  Only what the drivers'
output needs: groups with a TITLE attribute, contiguous numeric datasets, read-back of a whole
file into {"/group/name": ndarray}.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_CANDIDATES = ("libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so",
               "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so")

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 0, 0
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1

_NATIVE = {"float64": "H5T_NATIVE_DOUBLE_g", "float32": "H5T_NATIVE_FLOAT_g", "int8": "H5T_NATIVE_INT8_g",
           "uint8": "H5T_NATIVE_UINT8_g", "int16": "H5T_NATIVE_INT16_g", "uint16": "H5T_NATIVE_UINT16_g",
           "int32": "H5T_NATIVE_INT32_g", "uint32": "H5T_NATIVE_UINT32_g", "int64": "H5T_NATIVE_INT64_g",
           "uint64": "H5T_NATIVE_UINT64_g"}

_lib = None


class Hdf5Error(RuntimeError):
    pass


def _load():
    """the HDF5 library, or None"""
    global _lib
    if _lib is not None:
        return _lib or None
    names = [os.environ["LBMPM_HDF5_LIB"]] if os.environ.get("LBMPM_HDF5_LIB") else []
    found = ctypes.util.find_library("hdf5")
    names += ([found] if found else []) + list(_CANDIDATES)
    for n in names:
        try:
            L = C.CDLL(n)
        except OSError:
            continue
        if L.H5open() < 0:
            continue
        maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
        L.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
        L.version = (maj.value, mnr.value, rel.value)
        L.hid = C.c_int64 if (maj.value, mnr.value) >= (1, 10) else C.c_int      # hid_t grew to 64 bits in 1.10
        _declare(L)
        _lib = L
        return L
    _lib = False
    return None


def available():
    return _load() is not None


def _declare(L):
    hid, hsz = L.hid, C.c_uint64
    sig = {
        "H5Fcreate": (hid, [C.c_char_p, C.c_uint, hid, hid]), "H5Fopen": (hid, [C.c_char_p, C.c_uint, hid]),
        "H5Fclose": (C.c_int, [hid]),
        "H5Gcreate2": (hid, [hid, C.c_char_p, hid, hid, hid]), "H5Gopen2": (hid, [hid, C.c_char_p, hid]),
        "H5Gclose": (C.c_int, [hid]), "H5Gget_info": (C.c_int, [hid, C.c_void_p]),
        "H5Screate_simple": (hid, [C.c_int, C.POINTER(hsz), C.POINTER(hsz)]), "H5Screate": (hid, [C.c_int]),
        "H5Sclose": (C.c_int, [hid]), "H5Sget_simple_extent_ndims": (C.c_int, [hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid, C.POINTER(hsz), C.POINTER(hsz)]),
        "H5Dcreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), "H5Dopen2": (hid, [hid, C.c_char_p, hid]),
        "H5Dwrite": (C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
        "H5Dread": (C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
        "H5Dget_space": (hid, [hid]), "H5Dget_type": (hid, [hid]), "H5Dclose": (C.c_int, [hid]),
        "H5Tget_class": (C.c_int, [hid]), "H5Tget_size": (C.c_size_t, [hid]), "H5Tget_sign": (C.c_int, [hid]),
        "H5Tcopy": (hid, [hid]), "H5Tset_size": (C.c_int, [hid, C.c_size_t]), "H5Tclose": (C.c_int, [hid]),
        "H5Acreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid]), "H5Awrite": (C.c_int, [hid, hid, C.c_void_p]),
        "H5Aclose": (C.c_int, [hid]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid, C.c_char_p, C.c_int, C.c_int, hsz, C.c_char_p, C.c_size_t, hid]),
        "H5Eset_auto2": (C.c_int, [hid, C.c_void_p, C.c_void_p]),
        "H5Sselect_hyperslab": (C.c_int, [hid, C.c_int, C.POINTER(hsz), C.POINTER(hsz), C.POINTER(hsz), C.POINTER(hsz)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    L.H5Eset_auto2(0, None, None)          # errors come back as negative ids; no stack dumps on stderr


class _GInfo(C.Structure):          # H5G_info_t
    _fields_ = [("storage_type", C.c_int), ("nlinks", C.c_uint64), ("max_corder", C.c_int64), ("mounted", C.c_int)]


def _need():
    L = _load()
    if L is None:
        raise Hdf5Error("no HDF5 library found (set LBMPM_HDF5_LIB)")
    return L


def _ok(v, what):
    if v < 0:
        raise Hdf5Error("HDF5: %s failed" % what)
    return v


def _native(L, dtype):
    key = np.dtype(dtype).name
    if key not in _NATIVE:
        raise Hdf5Error("dtype %s is not supported by the HDF5 result writer" % key)
    return L.hid.in_dll(L, _NATIVE[key]).value


def _set_title(L, obj, title):
    raw = title.encode() + b"\0"
    t = _ok(L.H5Tcopy(L.hid.in_dll(L, "H5T_C_S1_g").value), "H5Tcopy")
    L.H5Tset_size(t, len(raw))
    s = _ok(L.H5Screate(H5S_SCALAR), "H5Screate")
    a = _ok(L.H5Acreate2(obj, b"TITLE", t, s, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2")
    L.H5Awrite(a, t, C.c_char_p(raw))
    L.H5Aclose(a); L.H5Sclose(s); L.H5Tclose(t)


def create(path, groups):
    """new file with the groups [(name, title), ...] under the root"""
    L = _need()
    f = _ok(L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), "H5Fcreate(%s)" % path)
    try:
        for name, title in groups:
            g = _ok(L.H5Gcreate2(f, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2(%s)" % name)
            _set_title(L, g, title)
            L.H5Gclose(g)
    finally:
        L.H5Fclose(f)


def write(path, name, array):
    """add the dataset `name` (absolute HDF5 path, its group must exist) to an existing file"""
    L = _need()
    a = np.ascontiguousarray(array)
    mem = _native(L, a.dtype)
    f = _ok(L.H5Fopen(path.encode(), H5F_ACC_RDWR, H5P_DEFAULT), "H5Fopen(%s)" % path)
    try:
        dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
        s = _ok(L.H5Screate_simple(a.ndim, dims, None) if a.ndim else L.H5Screate(H5S_SCALAR), "H5Screate_simple")
        d = L.H5Dcreate2(f, name.encode(), mem, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if d < 0:
            L.H5Sclose(s)
            raise Hdf5Error("HDF5: cannot create dataset %s in %s (exists already, or its group is missing)" % (name, path))
        rc = L.H5Dwrite(d, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p))
        L.H5Dclose(d); L.H5Sclose(s)
        _ok(rc, "H5Dwrite(%s)" % name)
    finally:
        L.H5Fclose(f)


def _members(L, f, group):
    g = _ok(L.H5Gopen2(f, group.encode(), H5P_DEFAULT), "H5Gopen2(%s)" % group)
    info = _GInfo()
    _ok(L.H5Gget_info(g, C.byref(info)), "H5Gget_info")
    L.H5Gclose(g)
    out = []
    for i in range(info.nlinks):
        n = L.H5Lget_name_by_idx(f, group.encode(), H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
        buf = C.create_string_buffer(int(n) + 1)
        L.H5Lget_name_by_idx(f, group.encode(), H5_INDEX_NAME, H5_ITER_INC, i, buf, int(n) + 1, H5P_DEFAULT)
        out.append(buf.value.decode())
    return out


def _read_dataset(L, d, z0=None, n=None):
    """the whole dataset, or its planes [z0, z0 + n) along the first axis (a hyperslab: the rest of the file is not read)"""
    s, t = L.H5Dget_space(d), L.H5Dget_type(d)
    nd = L.H5Sget_simple_extent_ndims(s)
    dims = (C.c_uint64 * max(nd, 1))()
    if nd > 0:
        L.H5Sget_simple_extent_dims(s, dims, None)
    mem_space = H5S_ALL
    if z0 is not None:
        if nd < 1 or z0 < 0 or n < 0 or z0 + n > dims[0]:
            L.H5Tclose(t); L.H5Sclose(s)
            raise Hdf5Error("planes [%d, %d) outside a dataset of shape %s" % (z0, z0 + n, tuple(dims[:nd])))
        start = (C.c_uint64 * nd)(*([z0] + [0] * (nd - 1)))
        count = (C.c_uint64 * nd)(*([n] + list(dims[1:nd])))
        _ok(L.H5Sselect_hyperslab(s, 0, start, None, count, None), "H5Sselect_hyperslab")
        mem_space = _ok(L.H5Screate_simple(nd, count, None), "H5Screate_simple")
        dims[0] = n
    cls, size = L.H5Tget_class(t), L.H5Tget_size(t)
    if cls == H5T_FLOAT:
        dt = np.dtype("f%d" % size)
    elif cls == H5T_INTEGER:
        dt = np.dtype(("i%d" if L.H5Tget_sign(t) else "u%d") % size)
    else:
        L.H5Tclose(t); L.H5Sclose(s)
        return None                               # strings, compounds: not part of the drivers' output
    out = np.empty(tuple(dims[:nd]), dtype=dt)
    rc = L.H5Dread(d, _native(L, dt), mem_space, s if z0 is not None else H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
    if mem_space != H5S_ALL:
        L.H5Sclose(mem_space)
    L.H5Tclose(t); L.H5Sclose(s)
    _ok(rc, "H5Dread")
    return out


def read_all(path):
    """{"/group/name": ndarray} of every numeric dataset in the file"""
    L = _need()
    f = _ok(L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), "H5Fopen(%s)" % path)
    out = {}

    def walk(group):
        for m in _members(L, f, group):
            full = (group.rstrip("/") + "/" + m)
            d = L.H5Dopen2(f, full.encode(), H5P_DEFAULT)
            if d >= 0:
                a = _read_dataset(L, d)
                L.H5Dclose(d)
                if a is not None:
                    out[full] = a
            else:
                walk(full)
    try:
        walk("/")
    finally:
        L.H5Fclose(f)
    return out


def read_planes(path, name, z0=None, n=None):
    """dataset `name` (absolute HDF5 path), or only its planes [z0, z0 + n) along the first axis"""
    L = _need()
    f = _ok(L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), "H5Fopen(%s)" % path)
    try:
        d = L.H5Dopen2(f, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(name)
        try:
            return _read_dataset(L, d, z0, n)
        finally:
            L.H5Dclose(d)
    finally:
        L.H5Fclose(f)
