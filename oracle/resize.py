"""NumPy oracle for the bilinear resize (test infrastructure).

The reference resizes with ``cv2.resize(im, None, None, fx=s, fy=s, interpolation=cv2.INTER_LINEAR)``
(code/detection/detection.py:63-78).  OpenCV is neither in the reference tree nor installed
(the reference pins no version); this restates the 8-bit INTER_LINEAR algorithm of OpenCV 3.x/4.x
(modules/imgproc/src/resize.cpp: the coordinate/weight tables of ``resize``, ``HResizeLinear`` and
``VResizeLinear`` with ``FixedPtCast<int, uchar, INTER_RESIZE_COEF_BITS*2>``, and the switch to the
2x2 area average for an exact factor-2 reduction).

PARITY UNPINNED: no cv2 here and no fixture in the reference; only the published algorithm pins it.
"""
import numpy as np


def dst_size(src_h, src_w, fx, fy):
    """cv::resize with dsize empty: saturate_cast<int>(size * f) = round half to even."""
    return int(np.rint(src_h * float(fy))), int(np.rint(src_w * float(fx)))


def _taps(dsize, scale, ssize):
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= ssize - 1
    f[hi], s[hi] = 0.0, ssize - 1
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    w1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, np.minimum(s + 1, ssize - 1), w0, w1


def resize_linear_u8(img, fx, fy):
    """img uint8 [H,W,C] -> uint8 [round(H*fy), round(W*fx), C]."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    dh, dw = dst_size(H, W, fx, fy)
    sx, sy = 1.0 / float(fx), 1.0 / float(fy)
    if sx == 2.0 and sy == 2.0 and H % 2 == 0 and W % 2 == 0:
        a = img.astype(np.int64)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _taps(dw, sx, W)
    y0, y1, b0, b1 = _taps(dh, sy, H)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]      # [H, dw, C]
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def resize(im, target_size, max_size):
    """detection.py:63-78: short side -> target_size unless the long side would exceed max_size."""
    im_size_min = min(im.shape[0], im.shape[1])
    im_size_max = max(im.shape[0], im.shape[1])
    im_scale = float(target_size) / float(im_size_min)
    if np.round(im_scale * im_size_max) > max_size:
        im_scale = float(max_size) / float(im_size_max)
    return resize_linear_u8(im, im_scale, im_scale), im_scale
