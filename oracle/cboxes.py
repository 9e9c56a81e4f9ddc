"""ctypes access to the C restatement of NMS (oracle/boxes_c.c); test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_LIB)
        _lib.oracle_nms_f32.restype = C.c_int
        _lib.oracle_nms_f32.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        _lib.oracle_nms_f64.restype = C.c_int
        _lib.oracle_nms_f64.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    return _lib


def nms(dets, thresh):
    """Same contract as oracle.boxes.nms; float32 or float64 by the dtype of ``dets``."""
    dets = np.ascontiguousarray(dets)
    n = dets.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int32)
    if dets.dtype == np.float64:
        k = lib().oracle_nms_f64(dets.ctypes.data, n, float(thresh), keep.ctypes.data)
    else:
        dets = dets.astype(np.float32, copy=False)
        k = lib().oracle_nms_f32(dets.ctypes.data, n, float(np.float32(thresh)), keep.ctypes.data)
    return keep[:k].tolist()
