"""HDF5 output of a propagation: ``save_output`` / ``save_datacube`` with the reference's file layout.

Stands in for paos/core/saveOutput.py:120-163 (``save_retval``), :166-223 (``save_output``) and :225-303
(``save_datacube``): one group ``S##`` per saved surface holding one dataset per key of the ``run()`` record
(``amplitude``, ``phase``, ``wfo``, ``wfe``, ``dx`` ... ``propagator``), the aperture object and the two ABCD
objects as sub-groups of their attributes, plus an ``info`` group; a data cube has one such tree per group tag
(``/<tag>/S##/<key>``).

The reference writes through h5py, which this image does not have; ``libhdf5`` itself is present
(``/opt/conda/lib/libhdf5.so.103``, 1.10.6), so the writer drives the C library through ctypes and encodes
values the way h5py does: Python ``str`` -> variable-length UTF-8 string scalar, ``int`` / ``float`` -> scalar
int64 / float64, ``tuple`` -> 1-D array, ``complex128`` arrays -> compound ``{r, i}`` of doubles, a list of
strings -> ``S10`` array of shape (len, 1).  **Parity note**: the layout follows the reference's code; it cannot be
compared with bytes written by h5py here (no h5py, no reference ``.h5`` fixture), and ``h5py_version`` in
``/info`` says so.  Without libhdf5 the functions raise ``RuntimeError`` (no fallback format).
"""
import ctypes
import ctypes.util
import datetime
import os

import numpy as np

_HID = ctypes.c_int64
_HSIZE = ctypes.c_uint64
_H5 = None


class _Lib:
    """The slice of the HDF5 C API (1.10) the writer and the test reader use."""

    F_ACC_RDONLY, F_ACC_TRUNC, F_ACC_EXCL = 0, 2, 4
    S_SCALAR = 0
    T_COMPOUND, T_STRING_CLASS, T_INTEGER_CLASS, T_FLOAT_CLASS, T_COMPOUND_CLASS = 6, 3, 0, 1, 6
    T_VARIABLE = ctypes.c_size_t(-1).value
    CSET_UTF8, STR_NULLPAD = 1, 1

    def __init__(self):
        names = [os.environ.get("PAOS_HDF5_LIB"), ctypes.util.find_library("hdf5"), "libhdf5.so",
                 "/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "libhdf5_serial.so"]
        lib = None
        for name in names:
            if not name:
                continue
            try:
                lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if lib is None:
            raise RuntimeError("libhdf5 not found (set PAOS_HDF5_LIB): HDF5 output is not available")
        self.lib = lib
        sig = {
            "H5open": (ctypes.c_int, []),
            "H5get_libversion": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint)] * 3),
            "H5Fcreate": (_HID, [ctypes.c_char_p, ctypes.c_uint, _HID, _HID]),
            "H5Fopen": (_HID, [ctypes.c_char_p, ctypes.c_uint, _HID]),
            "H5Fclose": (ctypes.c_int, [_HID]),
            "H5Gcreate2": (_HID, [_HID, ctypes.c_char_p, _HID, _HID, _HID]),
            "H5Gclose": (ctypes.c_int, [_HID]),
            "H5Screate": (_HID, [ctypes.c_int]),
            "H5Screate_simple": (_HID, [ctypes.c_int, ctypes.POINTER(_HSIZE), ctypes.POINTER(_HSIZE)]),
            "H5Sclose": (ctypes.c_int, [_HID]),
            "H5Sget_simple_extent_ndims": (ctypes.c_int, [_HID]),
            "H5Sget_simple_extent_dims": (ctypes.c_int, [_HID, ctypes.POINTER(_HSIZE), ctypes.POINTER(_HSIZE)]),
            "H5Dcreate2": (_HID, [_HID, ctypes.c_char_p, _HID, _HID, _HID, _HID, _HID]),
            "H5Dopen2": (_HID, [_HID, ctypes.c_char_p, _HID]),
            "H5Dwrite": (ctypes.c_int, [_HID, _HID, _HID, _HID, _HID, ctypes.c_void_p]),
            "H5Dread": (ctypes.c_int, [_HID, _HID, _HID, _HID, _HID, ctypes.c_void_p]),
            "H5Dget_space": (_HID, [_HID]),
            "H5Dget_type": (_HID, [_HID]),
            "H5Dclose": (ctypes.c_int, [_HID]),
            "H5Dvlen_reclaim": (ctypes.c_int, [_HID, _HID, _HID, ctypes.c_void_p]),
            "H5Tcopy": (_HID, [_HID]),
            "H5Tcreate": (_HID, [ctypes.c_int, ctypes.c_size_t]),
            "H5Tinsert": (ctypes.c_int, [_HID, ctypes.c_char_p, ctypes.c_size_t, _HID]),
            "H5Tset_size": (ctypes.c_int, [_HID, ctypes.c_size_t]),
            "H5Tset_cset": (ctypes.c_int, [_HID, ctypes.c_int]),
            "H5Tset_strpad": (ctypes.c_int, [_HID, ctypes.c_int]),
            "H5Tget_class": (ctypes.c_int, [_HID]),
            "H5Tget_size": (ctypes.c_size_t, [_HID]),
            "H5Tis_variable_str": (ctypes.c_int, [_HID]),
            "H5Tclose": (ctypes.c_int, [_HID]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.H5open() < 0:
            raise RuntimeError("H5open failed")
        # failures come back as exceptions with the name of the call (_ok); the library's own stack dump on stderr
        # adds nothing (h5py switches it off as well)
        lib.H5Eset_auto2.restype, lib.H5Eset_auto2.argtypes = ctypes.c_int, [_HID, ctypes.c_void_p, ctypes.c_void_p]
        lib.H5Eset_auto2(0, None, None)
        self.NATIVE_DOUBLE = _HID.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
        self.NATIVE_INT64 = _HID.in_dll(lib, "H5T_NATIVE_INT64_g").value
        self.C_S1 = _HID.in_dll(lib, "H5T_C_S1_g").value
        v = [ctypes.c_uint() for _ in range(3)]
        lib.H5get_libversion(*[ctypes.byref(x) for x in v])
        self.version = ".".join(str(x.value) for x in v)

    def __getattr__(self, name):
        return getattr(self.lib, name)


def _h5():
    global _H5
    if _H5 is None:
        _H5 = _Lib()
    return _H5


def hdf5_available():
    try:
        _h5()
        return True
    except RuntimeError:
        return False


def _ok(value, what):
    if value < 0:
        raise RuntimeError(f"HDF5: {what} failed")
    return value


class _Types:
    """Datatypes made once per file."""

    def __init__(self, h):
        self.h = h
        self.vstr = _ok(h.H5Tcopy(h.C_S1), "H5Tcopy")
        _ok(h.H5Tset_size(self.vstr, h.T_VARIABLE), "H5Tset_size")
        _ok(h.H5Tset_cset(self.vstr, h.CSET_UTF8), "H5Tset_cset")
        self.s10 = _ok(h.H5Tcopy(h.C_S1), "H5Tcopy")
        _ok(h.H5Tset_size(self.s10, 10), "H5Tset_size")
        _ok(h.H5Tset_strpad(self.s10, h.STR_NULLPAD), "H5Tset_strpad")
        self.c128 = _ok(h.H5Tcreate(h.T_COMPOUND, 16), "H5Tcreate")  # h5py's complex128: {r, i}
        _ok(h.H5Tinsert(self.c128, b"r", 0, h.NATIVE_DOUBLE), "H5Tinsert")
        _ok(h.H5Tinsert(self.c128, b"i", 8, h.NATIVE_DOUBLE), "H5Tinsert")

    def close(self):
        for t in (self.vstr, self.s10, self.c128):
            self.h.H5Tclose(t)


def _write_array(h, types, group, name, arr):
    arr = np.array(np.ma.getdata(arr), order="C", copy=True)  # a masked wfe map is stored as its data, like h5py stores it
    if arr.dtype == np.complex128:
        dtype = types.c128
    elif arr.dtype.kind == "f":
        arr = arr.astype(np.float64, copy=False)
        dtype = h.NATIVE_DOUBLE
    elif arr.dtype.kind in "iub":
        arr = arr.astype(np.int64, copy=False)
        dtype = h.NATIVE_INT64
    else:
        raise NameError(f"Data type not supported: {name} has dtype {arr.dtype}")
    if arr.ndim == 0:
        space = _ok(h.H5Screate(h.S_SCALAR), "H5Screate")
    else:
        dims = (_HSIZE * arr.ndim)(*arr.shape)
        space = _ok(h.H5Screate_simple(arr.ndim, dims, None), "H5Screate_simple")
    ds = _ok(h.H5Dcreate2(group, name.encode(), dtype, space, 0, 0, 0), f"H5Dcreate2({name})")
    try:
        _ok(h.H5Dwrite(ds, dtype, 0, 0, 0, arr.ctypes.data_as(ctypes.c_void_p)), f"H5Dwrite({name})")
    finally:
        h.H5Dclose(ds)
        h.H5Sclose(space)


def _write_string(h, types, group, name, text):
    space = _ok(h.H5Screate(h.S_SCALAR), "H5Screate")
    ds = _ok(h.H5Dcreate2(group, name.encode(), types.vstr, space, 0, 0, 0), f"H5Dcreate2({name})")
    try:
        buf = ctypes.c_char_p(text.encode("utf-8"))
        _ok(h.H5Dwrite(ds, types.vstr, 0, 0, 0, ctypes.byref(buf)), f"H5Dwrite({name})")
    finally:
        h.H5Dclose(ds)
        h.H5Sclose(space)


def _write_ascii_list(h, types, group, name, items):
    raw = b"".join(s.encode("ascii", "ignore")[:10].ljust(10, b"\0") for s in items)
    dims = (_HSIZE * 2)(len(items), 1)
    space = _ok(h.H5Screate_simple(2, dims, None), "H5Screate_simple")
    ds = _ok(h.H5Dcreate2(group, name.encode(), types.s10, space, 0, 0, 0), f"H5Dcreate2({name})")
    try:
        buf = ctypes.create_string_buffer(raw, len(raw))
        _ok(h.H5Dwrite(ds, types.s10, 0, 0, 0, buf), f"H5Dwrite({name})")
    finally:
        h.H5Dclose(ds)
        h.H5Sclose(space)


def _save_tree(h, types, tree, group):
    """saveOutput.py:43-80: dict -> sub-group, str / int / float / tuple / ndarray -> dataset, list of
    strings -> S10 column, None -> skipped; anything else is an error."""
    for key, data in tree.items():
        if isinstance(data, dict):
            sub = _ok(h.H5Gcreate2(group, str(key).encode(), 0, 0, 0), f"H5Gcreate2({key})")
            try:
                _save_tree(h, types, data, sub)
            finally:
                h.H5Gclose(sub)
        elif isinstance(data, str):
            _write_string(h, types, group, key, data)
        elif isinstance(data, (bool, int, float, np.integer, np.floating, tuple)):
            _write_array(h, types, group, key, np.asarray(data))
        elif isinstance(data, np.ndarray):
            _write_array(h, types, group, key, data)
        elif isinstance(data, list):
            _write_ascii_list(h, types, group, key, data)
        elif data is None:
            continue
        else:
            raise NameError(f"Data type not supported: {key} is a {type(data).__name__}")


def _object_fields(obj):
    """What the reference stores for an aperture or ABCD object is its ``__dict__`` (saveOutput.py:145-149): for the
    photutils apertures ``positions, a, b, theta`` / ``positions, w, h, theta``, for ABCD ``_ABCD, _cin, _cout``."""
    if hasattr(obj, "cin") and callable(obj):  # an ABCD (or the lazily multiplied product of a saved surface)
        return {"_ABCD": np.array(obj(), dtype=np.float64), "_cin": float(obj.cin), "_cout": float(obj.cout)}
    out = {"positions": np.asarray(obj.positions, dtype=np.float64)}
    for name in ("a", "b", "w", "h", "theta"):
        if hasattr(obj, name):
            out[name] = float(getattr(obj, name))
    return out


def _surface_tree(record, keys_to_keep):
    item = dict(record)
    if item.get("aperture") is not None:
        item["aperture"] = _object_fields(item["aperture"])
    for name in ("ABCDs", "ABCDt"):
        if name in item:
            item[name] = _object_fields(item[name])
    if keys_to_keep is not None:
        item = {k: v for k, v in item.items() if k in keys_to_keep}
    return item


def _save_info(h, types, file_name, out):
    from . import __version__

    attrs = {
        "file_name": file_name,
        "file_time": datetime.datetime.now().isoformat(),
        "creator": "paos_amd (MI355X-native propagation core behind the PAOS run / WFO API)",
        "program_name": "paos_amd",
        "program_version": __version__,
        "HDF5_Version": h.version,
        "h5py_version": "none: libhdf5 driven through ctypes (paos_amd/save_output.py)",
    }
    grp = _ok(h.H5Gcreate2(out, b"info", 0, 0, 0), "H5Gcreate2(info)")
    try:
        _save_tree(h, types, attrs, grp)
    finally:
        h.H5Gclose(grp)


def _save_retval(h, types, retval, keys_to_keep, out):
    for index in retval.keys():
        grp = _ok(h.H5Gcreate2(out, f"S{index:02d}".encode(), 0, 0, 0), "H5Gcreate2(surface)")
        try:
            _save_tree(h, types, _surface_tree(retval[index], keys_to_keep), grp)
        finally:
            h.H5Gclose(grp)


def _create(h, file_name, overwrite):
    if os.path.isfile(file_name):
        if not overwrite:
            raise OSError(f"{file_name} exists (the reference would append and fail on the existing /info group)")
        os.remove(file_name)
    return _ok(h.H5Fcreate(file_name.encode(), h.F_ACC_EXCL, 0, 0), f"H5Fcreate({file_name})")


def save_output(retval, file_name, keys_to_keep=None, overwrite=True):
    """``save_output`` of paos/core/saveOutput.py:166-223: ``/info`` and one ``/S##`` group per saved surface."""
    assert isinstance(retval, dict), "parameter retval must be a dict"
    assert isinstance(file_name, str), "parameter file_name must be a string"
    if keys_to_keep is not None:
        assert isinstance(keys_to_keep, list), "parameter keys_to_keep must be a list of strings"
    h = _h5()
    out = _create(h, file_name, overwrite)
    types = _Types(h)
    try:
        _save_info(h, types, file_name, out)
        _save_retval(h, types, retval, keys_to_keep, out)
    finally:
        types.close()
        h.H5Fclose(out)


def save_datacube(retval_list, file_name, group_names, keys_to_keep=None, overwrite=True):
    """``save_datacube`` of paos/core/saveOutput.py:225-303: ``/info`` and ``/<group name>/S##/<key>`` for each
    simulation of the list (e.g. one per wavelength)."""
    assert isinstance(retval_list, list), "parameter retval_list must be a list"
    assert isinstance(file_name, str), "parameter file_name must be a string"
    assert isinstance(group_names, list), "parameter group_names must be a list of strings"
    if keys_to_keep is not None:
        assert isinstance(keys_to_keep, list), "parameter keys_to_keep must be a list of strings"
    h = _h5()
    cube = _create(h, file_name, overwrite)
    types = _Types(h)
    try:
        _save_info(h, types, file_name, cube)
        for group_name, retval in zip(group_names, retval_list):
            grp = _ok(h.H5Gcreate2(cube, str(group_name).encode(), 0, 0, 0), f"H5Gcreate2({group_name})")
            try:
                _save_retval(h, types, retval, keys_to_keep, grp)
            finally:
                h.H5Gclose(grp)
    finally:
        types.close()
        h.H5Fclose(cube)


def read_dataset(file_name, path):
    """One dataset of a file back as a NumPy array / str (float64, int64, {r, i} complex, variable-length and
    fixed-length strings): what the tests use to check the writer, not a general reader."""
    h = _h5()
    f = _ok(h.H5Fopen(file_name.encode(), h.F_ACC_RDONLY, 0), f"H5Fopen({file_name})")
    try:
        ds = _ok(h.H5Dopen2(f, path.encode(), 0), f"H5Dopen2({path})")
        try:
            space, dtype = h.H5Dget_space(ds), h.H5Dget_type(ds)
            try:
                nd = h.H5Sget_simple_extent_ndims(space)
                dims = (_HSIZE * max(nd, 1))()
                if nd > 0:
                    h.H5Sget_simple_extent_dims(space, dims, None)
                shape = tuple(int(d) for d in dims[:nd])
                cls, size = h.H5Tget_class(dtype), h.H5Tget_size(dtype)
                if cls == h.T_STRING_CLASS and h.H5Tis_variable_str(dtype) > 0:
                    buf = ctypes.c_char_p()
                    _ok(h.H5Dread(ds, dtype, 0, 0, 0, ctypes.byref(buf)), "H5Dread")
                    text = buf.value.decode("utf-8")
                    h.H5Dvlen_reclaim(dtype, space, 0, ctypes.byref(buf))
                    return text
                if cls == h.T_STRING_CLASS:
                    out = np.empty(shape, dtype=f"S{size}")
                    _ok(h.H5Dread(ds, dtype, 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p)), "H5Dread")
                    return out
                if cls == h.T_COMPOUND_CLASS and size == 16:
                    out = np.empty(shape, dtype=np.complex128)
                elif cls == h.T_FLOAT_CLASS and size == 8:
                    out = np.empty(shape, dtype=np.float64)
                elif cls == h.T_INTEGER_CLASS and size == 8:
                    out = np.empty(shape, dtype=np.int64)
                else:
                    raise NameError(f"{path}: datatype class {cls} of {size} bytes is not read by this helper")
                _ok(h.H5Dread(ds, dtype, 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p)), "H5Dread")
                return out
            finally:
                h.H5Tclose(dtype)
                h.H5Sclose(space)
        finally:
            h.H5Dclose(ds)
    finally:
        h.H5Fclose(f)
