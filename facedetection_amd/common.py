"""Frame helpers with the names and call signatures of the reference's
code/common/Function.py, for the steps that sit between tracker and recogniser in its video
loop (code/data_processing/ProcessVideo.py:143-158).  The pixel work runs in the HIP library."""
import numpy as np
import torch

from . import ops


def stick_boxes(img_np, boxes):
    """Function.py:49-70: ``(img, crops)`` where crop i is the slice of ``img`` under box i after
    10 % padding and clamping to the frame.  ``boxes`` [n,>=4] are detector or tracker rows.
    The slice bounds are computed by the library (same arithmetic as the batched
    ``ops.crop_resize_pad``); the crops are views into ``img_np`` like the reference's.  The
    reference also draws a green rectangle into the frame for display; that is not done here,
    so the crops hold the untouched pixels."""
    boxes = np.asarray(boxes.cpu() if isinstance(boxes, torch.Tensor) else boxes)
    if boxes.shape[0] == 0:
        return img_np, []
    _, rects = ops.crop_rects(boxes, int(img_np.shape[0]), int(img_np.shape[1]))
    return img_np, [img_np[r[1]:r[3], r[0]:r[2]] for r in rects]


def resize_pad(img_np, des_size):
    """Function.py:95-113: letterbox one uint8 image [h,w,C] to ``des_size`` = (height, width)."""
    h, w = int(img_np.shape[0]), int(img_np.shape[1])
    tiles, _ = ops.crop_resize_pad(img_np, np.array([[0.0, 0.0, float(w), float(h)]]), des_size)
    return tiles[0]
