"""Bounding-box processing on the GPU, under the names and argument conventions
of the reference's code/detection/symbol/processing.py.

Inputs may be NumPy arrays (as in the reference) or torch tensors; NumPy in gives
NumPy out.  Every function runs a HIP kernel through the C ABI -- there is no
host implementation here.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib, runtime


def _is_np(x):
    return not isinstance(x, torch.Tensor)


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    """Anchor windows for each ratio x scale around a (0,0,base-1,base-1) box
    (reference: processing.py:184-248).  Returns float64 [R*S, 4]."""
    ratios = np.ascontiguousarray(ratios, dtype=np.float64).ravel()
    scales = np.ascontiguousarray(scales, dtype=np.float64).ravel()
    out = np.empty((ratios.size * scales.size, 4), dtype=np.float64)
    p = lambda a: a.ctypes.data_as(_lib.c_f64p)
    _lib.check(_lib.lib().fd_generate_anchors(float(base_size), p(ratios), ratios.size, p(scales), scales.size, p(out)))
    return out


def bbox_pred(boxes, box_deltas, is_train=False):
    """Apply [N, 4*C] deltas to [N, 4] boxes -> float64 [N, 4*C]
    (reference: processing.py:35-78, inference branch)."""
    if is_train:
        raise NotImplementedError("the training-time delta clamp (processing.py:56-64) is outside the detection path")
    as_np = _is_np(boxes)
    n = int(boxes.shape[0])
    cols = int(box_deltas.shape[1])
    if n == 0:
        return np.zeros((0, cols)) if as_np else torch.zeros((0, cols), dtype=torch.float64)
    if cols % 4 != 0 or int(box_deltas.shape[0]) != n or int(boxes.shape[1]) != 4:
        raise ValueError("bbox_pred: boxes [N,4] and box_deltas [N,4*C] expected")
    ctx = runtime.context()
    b = runtime.to_device(boxes, torch.float32)
    d = runtime.to_device(box_deltas, torch.float32)
    out = runtime.empty((n, cols), torch.float64)
    _lib.check(_lib.lib().fd_bbox_pred_f64(ctx.handle, runtime.ptr(b), runtime.ptr(d), n, cols // 4, 0, 0.0, 0.0,
                                           runtime.ptr(out), runtime.stream_ptr()))
    return out.cpu().numpy() if as_np else out


def clip_boxes(boxes, im_shape):
    """Clamp x to [0, im_shape[1]-1] and y to [0, im_shape[0]-1]
    (reference: processing.py:81-97).  NumPy input is modified in place, like the reference."""
    as_np = _is_np(boxes)
    n, cols = int(boxes.shape[0]), int(boxes.shape[1])
    if n == 0:
        return boxes
    ctx = runtime.context()
    t = runtime.to_device(boxes, torch.float64)
    _lib.check(_lib.lib().fd_clip_boxes_f64(ctx.handle, runtime.ptr(t), n, cols // 4, float(im_shape[0]),
                                            float(im_shape[1]), runtime.stream_ptr()))
    if as_np:
        boxes[...] = t.cpu().numpy().astype(boxes.dtype, copy=False)
        return boxes
    return t


def clip_pad(boxes, pad_shape):
    """Crop a [n, c, H, W] tensor to [n, c, h, w] (reference: processing.py:99-112); a view, no arithmetic."""
    h, w = pad_shape
    return boxes[:, :, :h, :w]


def nms(dets, thresh):
    """Greedy IoU suppression (reference: processing.py:114-148).

    dets [N,5] = (x1, y1, x2, y2, score) in float32 or float64; arithmetic runs in that
    dtype on the GPU.  Returns the kept row indices as a Python list, best score first.
    Equal scores are visited higher index first (the reference's order there is undefined).
    """
    n = int(dets.shape[0])
    if n == 0:
        return []
    if isinstance(dets, torch.Tensor):
        f64 = dets.dtype == torch.float64
    else:
        f64 = np.asarray(dets).dtype == np.float64
    ctx = runtime.context()
    d = runtime.to_device(dets, torch.float64 if f64 else torch.float32)
    if d.shape[1] != 5:
        raise ValueError("nms: dets must be [N,5]")
    keep = runtime.empty((n,), torch.int32)
    n_keep = runtime.empty((1,), torch.int32)
    fn = _lib.lib().fd_nms_f64 if f64 else _lib.lib().fd_nms_f32
    _lib.check(fn(ctx.handle, runtime.ptr(d), n, float(thresh), runtime.ptr(keep), runtime.ptr(n_keep),
                  runtime.stream_ptr()))
    k = int(n_keep.item())
    return keep[:k].cpu().tolist()
