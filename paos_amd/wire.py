"""Packed form of a work description (wavelengths + optical chains) for the one broadcast of the fan-out.

The reference ships work to its joblib workers by pickling (paos/core/pipeline.py:139-150).  Across GPUs
the work description travels as ONE flat byte blob through ``paos_comm_bcast_blob`` (RCCL or TCP), in a
small self-describing format instead of a pickle: it can only ever decode to the plain data an optical
chain is made of -- None, bool, int, float, str, list, dict, NumPy arrays, masked arrays and ``ABCD``
matrices (their 2 x 2 array and the two direction flags) -- never to code.

    blob  := value
    value := tag(1 byte) payload
      N none | T true | F false | i int64 | d float64 | s u32 len + utf-8 | b u32 len + bytes
      l u32 count + values | m u32 count + (key value)* | a dtype-str(u8 len) u8 ndim + u64 dims + raw C-order bytes
      M masked array: data array + mask array | A ABCD: 4 float64 (row-major) + cin + cout
All integers little-endian.  A 24-surface Ariel chain packs to ~6 KB; a 512-wavelength sweep to ~3 MB.
"""
import struct

import numpy as np

from .abcd import ABCD


def _pack(obj, out):
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)):
        out.append(b"i" + struct.pack("<q", int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"d" + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        raw = obj.encode("utf-8")
        out.append(b"s" + struct.pack("<I", len(raw)) + raw)
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b"b" + struct.pack("<I", len(obj)) + bytes(obj))
    elif isinstance(obj, ABCD):
        m = np.asarray(obj(), dtype=np.float64)
        out.append(b"A" + struct.pack("<6d", m[0, 0], m[0, 1], m[1, 0], m[1, 1], float(obj.cin), float(obj.cout)))
    elif isinstance(obj, np.ma.MaskedArray):
        out.append(b"M")
        _pack(np.ma.getdata(obj), out)
        _pack(np.ma.getmaskarray(obj), out)
    elif isinstance(obj, np.ndarray):
        arr = np.ascontiguousarray(obj)
        if arr.dtype.hasobject:
            raise TypeError("object arrays cannot travel")
        dt = arr.dtype.str.encode("ascii")
        out.append(b"a" + struct.pack("<B", len(dt)) + dt + struct.pack("<B", arr.ndim) +
                   struct.pack(f"<{arr.ndim}Q", *arr.shape) + arr.tobytes())
    elif isinstance(obj, (list, tuple)):
        out.append(b"l" + struct.pack("<I", len(obj)))
        for v in obj:
            _pack(v, out)
    elif isinstance(obj, dict):
        out.append(b"m" + struct.pack("<I", len(obj)))
        for k, v in obj.items():
            _pack(k, out)
            _pack(v, out)
    else:
        raise TypeError(f"cannot pack {type(obj).__name__} into a work description")


def dumps(obj):
    out = []
    _pack(obj, out)
    return b"".join(out)


def _unpack(buf, pos):
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b"N":
        return None, pos
    if tag == b"T":
        return True, pos
    if tag == b"F":
        return False, pos
    if tag == b"i":
        return struct.unpack_from("<q", buf, pos)[0], pos + 8
    if tag == b"d":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"s", b"b"):
        n = struct.unpack_from("<I", buf, pos)[0]
        raw = bytes(buf[pos + 4:pos + 4 + n])
        if len(raw) != n:
            raise ValueError("truncated work description")
        return (raw.decode("utf-8") if tag == b"s" else raw), pos + 4 + n
    if tag == b"A":
        a, b, c, d, cin, cout = struct.unpack_from("<6d", buf, pos)
        m = ABCD()
        m.ABCD = np.array([[a, b], [c, d]])
        m.cin, m.cout = np.float64(cin), np.float64(cout)
        return m, pos + 48
    if tag == b"M":
        data, pos = _unpack(buf, pos)
        mask, pos = _unpack(buf, pos)
        return np.ma.MaskedArray(data, mask=mask), pos
    if tag == b"a":
        n = buf[pos]
        dt = np.dtype(bytes(buf[pos + 1:pos + 1 + n]).decode("ascii"))
        if dt.hasobject:
            raise ValueError("object arrays cannot travel")
        pos += 1 + n
        ndim = buf[pos]
        shape = struct.unpack_from(f"<{ndim}Q", buf, pos + 1)
        pos += 1 + 8 * ndim
        count = int(np.prod(shape, dtype=np.int64)) if ndim else 1
        nbytes = count * dt.itemsize
        raw = bytes(buf[pos:pos + nbytes])
        if len(raw) != nbytes:
            raise ValueError("truncated work description")
        return np.frombuffer(raw, dtype=dt).reshape(shape).copy(), pos + nbytes
    if tag == b"l":
        n = struct.unpack_from("<I", buf, pos)[0]
        pos += 4
        items = []
        for _ in range(n):
            v, pos = _unpack(buf, pos)
            items.append(v)
        return items, pos
    if tag == b"m":
        n = struct.unpack_from("<I", buf, pos)[0]
        pos += 4
        d = {}
        for _ in range(n):
            k, pos = _unpack(buf, pos)
            v, pos = _unpack(buf, pos)
            d[k] = v
        return d, pos
    raise ValueError(f"unknown tag {tag!r} in a work description")


def loads(blob):
    buf = memoryview(bytes(blob))
    try:
        obj, pos = _unpack(buf, 0)
    except (struct.error, IndexError) as exc:
        raise ValueError(f"truncated work description ({exc})") from None
    if pos != len(buf):
        raise ValueError("trailing bytes after the work description")
    return obj
