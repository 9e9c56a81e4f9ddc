"""Operator-level entry points of the detection graph, one per native operator the
reference graph calls (symbol/faster_rcnn.py:98-103) plus the detector tail
(detection.py:47-58).  NumPy in -> NumPy out, torch in -> torch (device) out."""
import ctypes as C

import numpy as np
import torch

from . import _lib, runtime
from .symbol.config import config as _global_config


def make_proposal_cfg(feature_stride=16, scales=(4, 8, 16, 32), ratios=(0.5, 1, 2), rpn_pre_nms_top_n=None,
                      rpn_post_nms_top_n=None, threshold=None, rpn_min_size=None, box_f64=True):
    """fd_proposal_cfg from the keyword names of ``mx.symbol.contrib.Proposal``; values left
    at None are read from the global config, as symbol/faster_rcnn.py:100-102 does."""
    T = _global_config.TEST
    cfg = _lib.ProposalCfg()
    cfg.feat_stride = int(feature_stride)
    scales, ratios = list(scales), list(ratios)
    if not (0 < len(scales) <= 8 and 0 < len(ratios) <= 8):
        raise ValueError("between 1 and 8 scales and ratios are supported")
    cfg.num_scales, cfg.num_ratios = len(scales), len(ratios)
    for i, s in enumerate(scales):
        cfg.scales[i] = float(s)
    for i, r in enumerate(ratios):
        cfg.ratios[i] = float(r)
    cfg.pre_nms_top_n = int(T.RPN_PRE_NMS_TOP_N if rpn_pre_nms_top_n is None else rpn_pre_nms_top_n)
    cfg.post_nms_top_n = int(T.RPN_POST_NMS_TOP_N if rpn_post_nms_top_n is None else rpn_post_nms_top_n)
    cfg.nms_thresh = float(T.RPN_NMS_THRESH if threshold is None else threshold)
    cfg.min_size = float(T.RPN_MIN_SIZE if rpn_min_size is None else rpn_min_size)
    cfg.box_f64 = 1 if box_f64 else 0
    return cfg


def proposal(cls_prob, bbox_pred, im_info, output_score=False, return_counts=False, **kwargs):
    """``contrib.Proposal``: cls_prob [B,2A,H,W], bbox_pred [B,4A,H,W], im_info [B,3] ->
    rois [B*post_n, 5].  Raises ValueError on NaN input like the reference operator
    (proposal.py:47-51).  ``kwargs`` are those of :func:`make_proposal_cfg`."""
    as_np = not isinstance(cls_prob, torch.Tensor)
    cfg = make_proposal_cfg(**kwargs)
    if cfg.post_nms_top_n <= 0:
        raise ValueError("rpn_post_nms_top_n must be positive (the output has a fixed number of rows)")
    B, C2, H, W = (int(v) for v in cls_prob.shape)
    A = cfg.num_scales * cfg.num_ratios
    if C2 != 2 * A or tuple(int(v) for v in bbox_pred.shape[:2]) != (B, 4 * A):
        raise ValueError("proposal: cls_prob [B,%d,H,W] and bbox_pred [B,%d,H,W] expected" % (2 * A, 4 * A))
    ctx = runtime.context()
    cls_d = runtime.to_device(cls_prob, torch.float32)
    box_d = runtime.to_device(bbox_pred, torch.float32)
    if tuple(box_d.shape[2:]) != (H, W):          # clip_pad (processing.py:99-112)
        box_d = box_d[:, :, :H, :W].contiguous()
    info = np.ascontiguousarray(np.asarray(im_info.cpu() if isinstance(im_info, torch.Tensor) else im_info,
                                           dtype=np.float32).reshape(B, 3))
    rois = runtime.empty((B * cfg.post_nms_top_n, 5), torch.float32)
    scores = runtime.empty((B * cfg.post_nms_top_n,), torch.float32)
    counts = runtime.empty((B, 3), torch.int32)
    nan_flag = runtime.empty((1,), torch.int32)
    _lib.check(_lib.lib().fd_proposal(ctx.handle, C.byref(cfg), runtime.ptr(cls_d), runtime.ptr(box_d),
                                      info.ctypes.data_as(_lib.c_f32p), B, H, W, runtime.ptr(rois),
                                      runtime.ptr(scores), runtime.ptr(counts), runtime.ptr(nan_flag),
                                      runtime.stream_ptr()))
    if int(nan_flag.item()) != 0:
        raise ValueError("there is nan in input scores or bbox_deltas")
    out = [rois.cpu().numpy() if as_np else rois]
    if output_score:
        s = scores.reshape(-1, 1)
        out.append(s.cpu().numpy() if as_np else s)
    if return_counts:
        out.append(counts.cpu().numpy())
    return out[0] if len(out) == 1 else tuple(out)


def roi_pooling(data, rois, pooled_size=(7, 7), spatial_scale=0.0625):
    """``mx.symbol.ROIPooling``: data [B,C,H,W], rois [R,5] -> [R,C,ph,pw]."""
    as_np = not isinstance(data, torch.Tensor)
    B, Cc, H, W = (int(v) for v in data.shape)
    R = int(rois.shape[0])
    ph, pw = int(pooled_size[0]), int(pooled_size[1])
    ctx = runtime.context()
    d = runtime.to_device(data, torch.float32)
    r = runtime.to_device(rois, torch.float32)
    out = runtime.empty((R, Cc, ph, pw), torch.float32)
    _lib.check(_lib.lib().fd_roipool(ctx.handle, runtime.ptr(d), B, Cc, H, W, runtime.ptr(r), R, ph, pw,
                                     float(spatial_scale), runtime.ptr(out), runtime.stream_ptr()))
    return out.cpu().numpy() if as_np else out


def postprocess(rois, scores, bbox_deltas, im_shape, thresh=0.8, nms_thresh=0.3, do_round=True):
    """Detector tail (detection.py:47-58): rois [R,4], scores [R,2], bbox_deltas [R,8] -> dets [n,5] float32."""
    as_np = not isinstance(rois, torch.Tensor)
    R = int(rois.shape[0])
    if R == 0:
        return np.zeros((0, 5), np.float32) if as_np else torch.zeros((0, 5))
    ctx = runtime.context()
    r = runtime.to_device(rois, torch.float32)
    s = runtime.to_device(scores, torch.float32)
    d = runtime.to_device(bbox_deltas, torch.float32)
    dets = runtime.empty((R, 5), torch.float32)
    n = runtime.empty((1,), torch.int32)
    _lib.check(_lib.lib().fd_postprocess(ctx.handle, runtime.ptr(r), runtime.ptr(s), runtime.ptr(d), R,
                                         int(im_shape[0]), int(im_shape[1]), float(thresh), float(nms_thresh),
                                         1 if do_round else 0, runtime.ptr(dets), runtime.ptr(n),
                                         runtime.stream_ptr()))
    k = int(n.item())
    out = dets[:k]
    return out.cpu().numpy() if as_np else out


def conv2d(x, weight, stride=1, pad=0, bias=None, pre_scale=None, pre_shift=None, relu=False, residual=None,
           precision="f32"):
    """``mx.sym.Convolution`` fused with its neighbours (see fd_conv2d in facedet.h).

    x [B,H,W,Cin] channels-last, weight [Cout,Cin,kh,kw] (MXNet layout); returns y [B,Ho,Wo,Cout].
    y = relu?(conv(relu(x*pre_scale+pre_shift)?) + bias?) + residual?"""
    as_np = not isinstance(x, torch.Tensor)
    B, H, W, Cin = (int(v) for v in x.shape)
    Cout, Cin_w, kh, kw = (int(v) for v in weight.shape)
    if Cin_w != Cin:
        raise ValueError("conv2d: weight expects %d input channels, x has %d" % (Cin_w, Cin))
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    ctx = runtime.context()
    dev = lambda a: None if a is None else runtime.to_device(a, torch.float32)
    xd, wd, bd, sd, hd, rd = dev(x), dev(weight), dev(bias), dev(pre_scale), dev(pre_shift), dev(residual)
    y = runtime.empty((B, Ho, Wo, Cout), torch.float32)
    if precision in ("winograd", "winograd4"):
        # float32, Winograd F(2x2,3x3) or F(4x4,3x3): the 3x3 / stride 1 / pad 1 case without prologue or residual
        if (kh, kw, int(stride), int(pad)) != (3, 3, 1, 1) or sd is not None or hd is not None or rd is not None:
            raise ValueError("conv2d: the Winograd form is the plain 3x3, stride 1, pad 1 convolution")
        _lib.check(_lib.lib().fd_conv2d_winograd(ctx.handle, runtime.ptr(xd), B, H, W, Cin, runtime.ptr(wd), Cout,
                                                  runtime.ptr(bd), 1 if relu else 0, 4 if precision == "winograd4" else 2,
                                                  runtime.ptr(y), runtime.stream_ptr()))
        return y.cpu().numpy() if as_np else y
    fn = {"f32": _lib.lib().fd_conv2d, "f16x3": _lib.lib().fd_conv2d_f16x3, "bf16x3": _lib.lib().fd_conv2d_bf16x3}[precision]
    _lib.check(fn(ctx.handle, runtime.ptr(xd), B, H, W, Cin, runtime.ptr(wd), Cout, kh, kw,
                  int(stride), int(pad), runtime.ptr(bd), runtime.ptr(sd), runtime.ptr(hd),
                  1 if relu else 0, runtime.ptr(rd), runtime.ptr(y), runtime.stream_ptr()))
    return y.cpu().numpy() if as_np else y


def resize_bilinear(im, fx, fy):
    """``cv2.resize(im, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR)`` for uint8 images
    [H,W,C] or batches [B,H,W,C] (C <= 4)."""
    as_np = not isinstance(im, torch.Tensor)
    single = im.ndim == 3
    t = runtime.to_device(im, torch.uint8)
    if single:
        t = t.unsqueeze(0)
    B, H, W, Cc = (int(v) for v in t.shape)
    dh, dw = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.lib().fd_resize_dims(H, W, float(fx), float(fy), C.byref(dh), C.byref(dw)))
    out = runtime.empty((B, dh.value, dw.value, Cc), torch.uint8)
    ctx = runtime.context()
    _lib.check(_lib.lib().fd_resize_u8(ctx.handle, runtime.ptr(t), B, H, W, Cc, float(fx), float(fy),
                                       runtime.ptr(out), dh.value, dw.value, runtime.stream_ptr()))
    if single:
        out = out[0]
    return out.cpu().numpy() if as_np else out


def crop_resize_pad(frame, boxes, des_size=(112, 112), tiles=True):
    """``Function.stick_boxes`` + ``Function.resize_pad`` (code/common/Function.py:49-70,95-113) for
    all boxes of one uint8 frame [H,W,C] in one launch.  ``boxes`` [n,>=4] (x1,y1,x2,y2,...):
    detections or tracker rows.  Returns (tiles uint8 [n,des_h,des_w,C], rects int32 [n,4]) as
    NumPy arrays for a NumPy frame, device tensors otherwise; ``tiles=False`` computes the
    rectangles only."""
    as_np = not isinstance(frame, torch.Tensor)
    f = runtime.to_device(frame, torch.uint8)
    H, W, Cc = (int(v) for v in f.shape)
    b = runtime.to_device(boxes if isinstance(boxes, torch.Tensor) else np.asarray(boxes, dtype=np.float64),
                          torch.float64)
    if b.dim() != 2 or b.shape[1] < 4:
        raise ValueError("boxes must be [n, >=4]")
    n = int(b.shape[0])
    dh, dw = int(des_size[0]), int(des_size[1])
    out = runtime.empty((n, dh, dw, Cc), torch.uint8) if tiles else None
    rects = runtime.empty((n, 4), torch.int32)
    ctx = runtime.context()
    _lib.check(_lib.lib().fd_crop_resize_pad(ctx.handle, runtime.ptr(f), H, W, Cc, runtime.ptr(b), int(b.shape[1]), n,
                                             dh, dw, runtime.ptr(out) if tiles else None, runtime.ptr(rects),
                                             runtime.stream_ptr()))
    if as_np:
        return (out.cpu().numpy() if tiles else None), rects.cpu().numpy()
    return out, rects


def crop_rects(boxes, height, width):
    """The padded, clamped slice bounds of ``Function.stick_boxes`` (code/common/Function.py:53-64)
    for boxes [n,>=4] in a frame of the given size: (None, int32 [n,4] NumPy array)."""
    b = runtime.to_device(boxes if isinstance(boxes, torch.Tensor) else np.asarray(boxes, dtype=np.float64),
                          torch.float64)
    n = int(b.shape[0])
    rects = runtime.empty((n, 4), torch.int32)
    ctx = runtime.context()
    _lib.check(_lib.lib().fd_crop_resize_pad(ctx.handle, None, int(height), int(width), 3, runtime.ptr(b),
                                             int(b.shape[1]), n, 1, 1, None, runtime.ptr(rects),
                                             runtime.stream_ptr()))
    return None, rects.cpu().numpy()
