"""NumPy oracle for the face-crop hand-off (test infrastructure).

Follows the reference's ``stick_boxes`` (code/common/Function.py:49-70: truncate the box to
integers, pad each side by round(10 % of the side), clamp to the frame, slice) and
``resize_pad`` (code/common/Function.py:95-113: scale the crop to touch the destination in
one direction with cv2.resize INTER_LINEAR, then centre it with black borders).

The reference also paints a green rectangle into the frame *before* slicing, so its crops
carry that overlay; this restatement (like the product) leaves the pixels untouched.

PARITY UNPINNED for the resampling: cv2 is absent, see oracle/resize.py.
"""
import numpy as np

from . import resize as oresize


def stick_rects(boxes, height, width):
    """int64 [n,4] (x1, y1, x2, y2), half-open slice bounds as in Function.py:53-64."""
    boxes = np.asarray(boxes)
    out = np.zeros((boxes.shape[0], 4), dtype=np.int64)
    for i in range(boxes.shape[0]):
        x1, y1, x2, y2 = (int(v) for v in boxes[i, :4])
        pw = int(np.round((x2 - x1) * 0.1))
        ph = int(np.round((y2 - y1) * 0.1))
        out[i] = (max(0, x1 - pw), max(0, y1 - ph), min(width, x2 + pw), min(height, y2 + ph))
    return out


def stick_boxes(img, boxes):
    rects = stick_rects(boxes, img.shape[0], img.shape[1])
    return img, [img[r[1]:r[3], r[0]:r[2]] for r in rects]


def resize_pad(img, des_size):
    """uint8 [h,w,C] -> uint8 [des_h,des_w,C] (Function.py:95-113)."""
    ratio_src = img.shape[0] / img.shape[1]
    ratio_des = des_size[0] / des_size[1]
    if ratio_src > ratio_des:
        scale = des_size[0] / img.shape[0]
    else:
        scale = des_size[1] / img.shape[1]
    small = oresize.resize_linear_u8(img, scale, scale)
    out = np.zeros((des_size[0], des_size[1], img.shape[2]), dtype=np.uint8)
    if ratio_src > ratio_des:
        delta = des_size[1] - small.shape[1]
        top, left = 0, delta // 2
    else:
        delta = des_size[0] - small.shape[0]
        top, left = delta // 2, 0
    out[top:top + small.shape[0], left:left + small.shape[1]] = small
    return out


def crop_resize_pad(img, boxes, des_size):
    """stick_boxes followed by resize_pad on every crop: uint8 [n,des_h,des_w,C] and the rects.
    A degenerate crop (no pixels) yields a black tile."""
    rects = stick_rects(boxes, img.shape[0], img.shape[1])
    out = np.zeros((len(rects), des_size[0], des_size[1], img.shape[2]), dtype=np.uint8)
    for i, r in enumerate(rects):
        if r[2] > r[0] and r[3] > r[1]:
            out[i] = resize_pad(img[r[1]:r[3], r[0]:r[2]], des_size)
    return out, rects
