"""Reader (and writer, for tests) of MXNet ``.params`` files -- the container
``mx.model.load_checkpoint(prefix, epoch)`` reads at the reference's detection.py:28
(``prefix-%04d.params``; the accompanying ``-symbol.json`` is not needed because the
graph is fixed: faster_rcnn50).

Format (``mx.nd.save`` of a dict; restated from the MXNet 0.x/1.x sources, nothing else is
available offline -- no real checkpoint could be tested here):

    uint64 0x112 (list magic), uint64 reserved
    uint64 n_arrays, then each array:
        uint32 magic   0xF993FAC9 (V2) / 0xF993FACA (V3)  -> int32 storage type (0 = dense),
                       shape = uint32 (V2) or int32 (V3) ndim + int64 dims
                       0xF993FAC8 (V1)                    -> shape = uint32 ndim + int64 dims
                       anything else: the word IS ndim (oldest files), uint32 dims follow
        int32 dev_type, int32 dev_id, int32 type_flag (0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64)
        raw little-endian data
    uint64 n_names, then each: uint64 length + bytes   ("arg:conv0_weight", "aux:bn0_moving_mean", ...)

The V1 magic was introduced together with 64-bit shape dimensions, so V1 records are read with
int64 dims first; because that cannot be checked against a real file here, a record that does
not validate that way is re-read with uint32 dims.  A candidate layout is accepted only if every
dimension is positive, the dtype flag is known, the payload fits the file, and what follows the
payload is the start of another record (or, after the last array, the name table).

``load_params`` returns one dict with the ``arg:`` / ``aux:`` prefixes stripped, i.e. the
merged ``arg_params`` + ``aux_params`` the detector wants.
"""
import struct

import numpy as np

LIST_MAGIC = 0x112
V1, V2, V3 = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
FLAGS = {np.dtype(v): k for k, v in DTYPES.items()}
MAX_NDIM = 32


class _Reader:
    def __init__(self, data, offset=0):
        self.b, self.o = data, offset

    def take(self, fmt):
        size = struct.calcsize("<" + fmt)
        if self.o + size > len(self.b):
            raise ValueError("truncated .params file")
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += size
        return v if len(v) > 1 else v[0]

    def raw(self, n):
        if n < 0 or self.o + n > len(self.b):
            raise ValueError("truncated .params file")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def peek(self, fmt):
        size = struct.calcsize("<" + fmt)
        if self.o + size > len(self.b):
            return None
        return struct.unpack_from("<" + fmt, self.b, self.o)[0]


def _follows_ok(r, last, n_arrays):
    """Does the file continue plausibly at r.o: another array record, or the name table?"""
    if last:
        n_names = r.peek("Q")
        return n_names is not None and n_names in (0, n_arrays)
    word = r.peek("I")
    return word is not None and (word in (V1, V2, V3) or word <= MAX_NDIM)


def _read_body(r, shape):
    """Context, dtype flag and payload of a dense record whose shape has been read."""
    r.take("ii")                       # context the array was saved from
    flag = r.take("i")
    if flag not in DTYPES:
        raise ValueError("unknown dtype flag %d" % flag)
    dt = np.dtype(DTYPES[flag]).newbyteorder("<")
    count = 1
    for d in shape:
        if d <= 0:
            raise ValueError("non-positive dimension %d" % d)
        count *= int(d)
    if count * dt.itemsize > len(r.b) - r.o:
        raise ValueError("truncated .params file (array of %d bytes, %d left)" % (count * dt.itemsize, len(r.b) - r.o))
    return np.frombuffer(r.raw(count * dt.itemsize), dtype=dt, count=count).reshape(shape).copy()


def _read_array(r, last, n_arrays):
    magic = r.take("I")
    if magic in (V2, V3):
        stype = r.take("i")
        if stype != 0:
            raise ValueError("sparse NDArrays (storage type %d) are not supported" % stype)
        ndim = r.take("i" if magic == V3 else "I")
        if ndim < 0 or (ndim == 0 and magic == V2):
            return None                # "none" array: nothing else is stored for it
        if ndim > MAX_NDIM:
            raise ValueError("not an MXNet NDArray record (ndim %d)" % ndim)
        return _read_body(r, [r.take("q") for _ in range(ndim)])
    ndim = r.take("I") if magic == V1 else magic
    if ndim > MAX_NDIM:
        raise ValueError("not an MXNet NDArray record (ndim %d)" % ndim)
    if ndim == 0:
        return None
    start, errors = r.o, []
    for dim_fmt in (("q", "I") if magic == V1 else ("I",)):
        t = _Reader(r.b, start)
        try:
            a = _read_body(t, [t.take(dim_fmt) for _ in range(ndim)])
            if not _follows_ok(t, last, n_arrays):
                raise ValueError("record is not followed by another record or the name table")
        except ValueError as e:
            errors.append("%s dims: %s" % ("int64" if dim_fmt == "q" else "uint32", e))
            continue
        r.o = t.o
        return a
    raise ValueError("unreadable NDArray record (" + "; ".join(errors) + ")")


def load_ndarray_dict(path):
    """name -> ndarray for every array of an ``mx.nd.save`` file (names keep their prefixes)."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    magic, _ = r.take("QQ")
    if magic != LIST_MAGIC:
        raise ValueError("%s is not an MXNet NDArray list file (magic 0x%x)" % (path, magic))
    n = r.take("Q")
    if n > len(r.b):
        raise ValueError("%s: implausible array count %d" % (path, n))
    arrays = [_read_array(r, i == n - 1, n) for i in range(n)]
    n_names = r.take("Q")
    if n_names not in (0, n):
        raise ValueError("name count %d does not match array count %d" % (n_names, n))
    names = [r.raw(r.take("Q")).decode("utf-8") for _ in range(n_names)]
    if n_names == 0:
        names = [str(i) for i in range(n)]
    return {k: v for k, v in zip(names, arrays) if v is not None}


def load_params(path):
    """Merged arg + aux parameters as float32, keyed by bare MXNet names."""
    out = {}
    for k, v in load_ndarray_dict(path).items():
        name = k.split(":", 1)[1] if k.startswith(("arg:", "aux:")) else k
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def save_params(path, arg_params, aux_params=None, version=2, v1_dims="q", n_names=None):
    """Writes the same container (for round-trip tests and for exporting synthetic checkpoints).
    ``version``: 0 legacy (no magic, uint32 dims), 1 (V1 magic, int64 dims; ``v1_dims="I"`` writes
    the 32-bit variant the reader also accepts), 2, 3.  ``n_names`` overrides the stored name count
    (negative tests)."""
    items = [("arg:" + k, v) for k, v in arg_params.items()]
    items += [("aux:" + k, v) for k, v in (aux_params or {}).items()]
    with open(path, "wb") as f:
        f.write(struct.pack("<QQQ", LIST_MAGIC, 0, len(items)))
        for _, v in items:
            a = np.ascontiguousarray(v)
            if version >= 2:
                f.write(struct.pack("<Ii", V2 if version == 2 else V3, 0))
                f.write(struct.pack("<i" if version == 3 else "<I", a.ndim))
                f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            elif version == 1:
                f.write(struct.pack("<II", V1, a.ndim))
                f.write(struct.pack("<%d%s" % (a.ndim, v1_dims), *a.shape))
            else:
                f.write(struct.pack("<I", a.ndim))
                f.write(struct.pack("<%dI" % a.ndim, *a.shape))
            f.write(struct.pack("<iii", 1, 0, FLAGS[a.dtype]))
            f.write(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
        names = items if n_names is None else items[:n_names]
        f.write(struct.pack("<Q", len(items) if n_names is None else n_names))
        for k, _ in names:
            kb = k.encode("utf-8")
            f.write(struct.pack("<Q", len(kb)) + kb)
