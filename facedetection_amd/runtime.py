"""Device plumbing: one library context per GPU, PyTorch-ROCm tensors as buffers.

torch is used for exactly three things here: allocating device memory, naming
the current HIP stream, and host<->device copies.  No torch operator takes part
in any computation of the detection path.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib

_contexts = {}
_lock = threading.Lock()


class Context:
    """Owns one ``fd_ctx`` (the library's scratch memory on one GPU)."""

    def __init__(self, device_index):
        self.device_index = int(device_index)
        handle = C.c_void_p()
        _lib.check(_lib.lib().fd_ctx_create(self.device_index, C.byref(handle)))
        self.handle = handle

    def scratch_bytes(self):
        return int(_lib.lib().fd_ctx_scratch_bytes(self.handle))

    def close(self):
        if self.handle:
            _lib.lib().fd_ctx_destroy(self.handle)
            self.handle = None


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("facedetection_amd needs a ROCm GPU (MI355X); none is visible and there is no CPU fallback")


def device_index(device=None):
    require_gpu()
    if device is None:
        return torch.cuda.current_device()
    if isinstance(device, int):
        return device
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def context(device=None):
    """The per-process context of ``device`` (created on first use)."""
    idx = device_index(device)
    with _lock:
        ctx = _contexts.get(idx)
        if ctx is None:
            with torch.cuda.device(idx):
                ctx = Context(idx)
            _contexts[idx] = ctx
    return ctx


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on ``device`` as an integer."""
    return torch.cuda.current_stream(device_index(device)).cuda_stream


def to_device(array, dtype, device=None):
    """Contiguous device tensor holding ``array`` (NumPy array or torch tensor)."""
    idx = device_index(device)
    if isinstance(array, torch.Tensor):
        return array.to(device="cuda:%d" % idx, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(array)).to(device="cuda:%d" % idx, dtype=dtype)


def empty(shape, dtype, device=None):
    return torch.empty(shape, dtype=dtype, device="cuda:%d" % device_index(device))


def ptr(tensor):
    return C.c_void_p(tensor.data_ptr()) if tensor is not None else C.c_void_p(0)
