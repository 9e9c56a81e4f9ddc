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
                       0xF993FAC8 (V1)                    -> shape = uint32 ndim + uint32 dims
                       anything else: the word IS ndim (oldest files), uint32 dims follow
        int32 dev_type, int32 dev_id, int32 type_flag (0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64)
        raw little-endian data
    uint64 n_names, then each: uint64 length + bytes   ("arg:conv0_weight", "aux:bn0_moving_mean", ...)

``load_params`` returns one dict with the ``arg:`` / ``aux:`` prefixes stripped, i.e. the
merged ``arg_params`` + ``aux_params`` the detector wants.
"""
import struct

import numpy as np

LIST_MAGIC = 0x112
V1, V2, V3 = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
FLAGS = {np.dtype(v): k for k, v in DTYPES.items()}


class _Reader:
    def __init__(self, data):
        self.b, self.o = data, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def raw(self, n):
        if self.o + n > len(self.b):
            raise ValueError("truncated .params file")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v


def _read_array(r):
    magic = r.take("I")
    if magic in (V2, V3):
        stype = r.take("i")
        if stype != 0:
            raise ValueError("sparse NDArrays (storage type %d) are not supported" % stype)
        ndim = r.take("i" if magic == V3 else "I")
        if ndim < 0:
            return None
        shape = [r.take("q") for _ in range(ndim)]
    else:
        ndim = r.take("I") if magic == V1 else magic
        if ndim > 32:
            raise ValueError("not an MXNet NDArray record (ndim %d)" % ndim)
        shape = [r.take("I") for _ in range(ndim)]
    if ndim == 0:
        return None
    r.take("ii")                       # context the array was saved from
    flag = r.take("i")
    if flag not in DTYPES:
        raise ValueError("unknown dtype flag %d" % flag)
    dt = np.dtype(DTYPES[flag]).newbyteorder("<")
    count = int(np.prod(shape))
    return np.frombuffer(r.raw(count * dt.itemsize), dtype=dt, count=count).reshape(shape).copy()


def load_ndarray_dict(path):
    """name -> ndarray for every array of an ``mx.nd.save`` file (names keep their prefixes)."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    magic, _ = r.take("QQ")
    if magic != LIST_MAGIC:
        raise ValueError("%s is not an MXNet NDArray list file (magic 0x%x)" % (path, magic))
    n = r.take("Q")
    arrays = [_read_array(r) for _ in range(n)]
    n_names = r.take("Q")
    names = [r.raw(r.take("Q")).decode("utf-8") for _ in range(n_names)]
    if n_names not in (0, n):
        raise ValueError("name count %d does not match array count %d" % (n_names, n))
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


def save_params(path, arg_params, aux_params=None, version=2):
    """Writes the same container (for round-trip tests and for exporting synthetic checkpoints)."""
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
            else:
                if version == 1:
                    f.write(struct.pack("<I", V1))
                f.write(struct.pack("<I", a.ndim))
                f.write(struct.pack("<%dI" % a.ndim, *a.shape))
            f.write(struct.pack("<iii", 1, 0, FLAGS[a.dtype]))
            f.write(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
        f.write(struct.pack("<Q", len(items)))
        for k, _ in items:
            kb = k.encode("utf-8")
            f.write(struct.pack("<Q", len(kb)) + kb)
