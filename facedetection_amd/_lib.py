"""ctypes binding of libfacedet.so (the C ABI declared in include/facedet.h).

Loading is lazy and strict: the first use of ``lib()`` loads the in-tree shared
library or raises ``RuntimeError`` -- there is no CPU fallback to fall through to.
"""
import ctypes as C
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
# FD_LIBRARY selects another build of the library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("FD_LIBRARY") or os.path.join(_PKG, "libfacedet.so")

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


class ProposalCfg(C.Structure):
    _fields_ = [
        ("feat_stride", C.c_int),
        ("num_scales", C.c_int),
        ("scales", C.c_float * 8),
        ("num_ratios", C.c_int),
        ("ratios", C.c_float * 8),
        ("pre_nms_top_n", C.c_int),
        ("post_nms_top_n", C.c_int),
        ("nms_thresh", C.c_float),
        ("min_size", C.c_float),
        ("box_f64", C.c_int),
    ]


class ProfileRow(C.Structure):
    _fields_ = [
        ("kernel", C.c_char * 64),
        ("launches", C.c_int64),
        ("ms", C.c_double),
        ("flops", C.c_double),
        ("bytes", C.c_double),
    ]


class Param(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("data_host", C.c_void_p),
        ("numel", C.c_int64),
    ]


# name -> (restype, argtypes); every symbol include/facedet.h declares
SIGNATURES = {
    "fd_version": (C.c_int, []),
    "fd_build_id": (C.c_char_p, []),
    "fd_last_error": (C.c_char_p, []),
    "fd_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "fd_ctx_destroy": (None, [vp]),
    "fd_ctx_scratch_bytes": (C.c_int64, [vp]),
    "fd_profile_begin": (C.c_int, []),
    "fd_profile_begin_only": (C.c_int, [C.c_char_p]),
    "fd_profile_end": (C.c_int, [C.POINTER(ProfileRow), C.c_int, C.POINTER(C.c_int)]),
    "fd_generate_anchors": (C.c_int, [C.c_double, c_f64p, C.c_int, c_f64p, C.c_int, c_f64p]),
    "fd_bbox_pred_f64": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp, vp]),
    "fd_clip_boxes_f64": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, vp]),
    "fd_nms_f32": (C.c_int, [vp, vp, C.c_int, C.c_float, vp, vp, vp]),
    "fd_nms_f32_noalloc": (C.c_int, [vp, C.c_int, C.c_float, vp, vp, vp]),
    "fd_nms_f64": (C.c_int, [vp, vp, C.c_int, C.c_double, vp, vp, vp]),
    "fd_proposal": (C.c_int, [vp, C.POINTER(ProposalCfg), vp, vp, c_f32p, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "fd_roipool": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]),
    "fd_postprocess": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, vp, vp, vp]),
    "fd_resize_dims": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, c_i32p, c_i32p]),
    "fd_resize_u8": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp, C.c_int,
                               C.c_int, vp]),
    "fd_crop_resize_pad": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp,
                                     vp]),
    "fd_resize_cvt_u8": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, vp,
                                   C.c_int, C.c_int, vp]),
    "fd_detect_scaled": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp, vp, vp, vp,
                                   vp]),
    "fd_conv2d": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            vp, vp, vp, C.c_int, vp, vp, vp]),
    "fd_conv2d_bf16x3": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, vp, vp, vp, C.c_int, vp, vp, vp]),
    "fd_conv2d_f16x3": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, vp, vp, vp, C.c_int, vp, vp, vp]),
    "fd_conv2d_winograd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]),
    "fd_detector_create": (C.c_int, [vp, C.POINTER(Param), C.c_int, C.POINTER(ProposalCfg), C.POINTER(vp)]),
    "fd_detector_destroy": (None, [vp]),
    "fd_detector_set_precision": (C.c_int, [vp, C.c_int]),
    "fd_detector_set_proposal": (C.c_int, [vp, C.POINTER(ProposalCfg)]),
    "fd_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp]),
    "fd_detect": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, vp, vp, vp, vp]),
    "fd_detector_status": (C.c_int, [vp, vp, vp]),
    "fd_detect_graph_create": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, vp,
                                         vp, vp, vp, C.POINTER(vp)]),
    "fd_detect_graph_launch": (C.c_int, [vp, vp]),
    "fd_detect_graph_destroy": (None, [vp]),
    "fd_detector_keep_tensors": (C.c_int, [vp, C.c_int]),
    "fd_detector_tensor": (C.c_int, [vp, C.c_char_p, vp, C.c_int64, C.POINTER(C.c_int64), vp]),
}

_lock = threading.Lock()
_lib = None


class FaceDetError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libfacedet error %d: %s" % (code, message))
        self.code = code


def lib():
    """The loaded library with argument types set; raises if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "%s is missing: build it with `python -m facedetection_amd._build` "
                    "(there is no CPU fallback)" % LIB_PATH)
            # torch ships its own HIP runtime; it has to be in the process before this library
            # so that both resolve to the same libamdhip64 (two runtimes in one process do not
            # see each other's devices)
            import torch  # noqa: F401
            handle = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)     # AttributeError if the symbol is not exported
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().fd_last_error()
        raise FaceDetError(rc, msg.decode("utf-8", "replace") if msg else "")


def declared_symbols(header_path=None):
    """Function names declared in include/facedet.h (used by the CPU-side ABI test)."""
    import re
    if header_path is None:
        header_path = os.path.join(os.path.dirname(_PKG), "include", "facedet.h")
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", text)))
