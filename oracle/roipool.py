"""NumPy oracle for ROI max pooling (test infrastructure).

The reference calls MXNet's native ``mx.symbol.ROIPooling`` (symbol/faster_rcnn.py:103);
MXNet is not in the reference tree nor installed here (version unpinned by the reference,
API level >= 1.3), so this restates the operator's published algorithm
(src/operator/roi_pooling.cc of Apache MXNet 1.x, itself the Fast R-CNN Caffe layer):

  * roi corners are multiplied by spatial_scale and rounded half away from zero;
  * roi height/width = max(end - start + 1, 1); bin size = roi size / pooled size (float32);
  * bin (ph, pw) spans rows floor(ph*bin_h) .. ceil((ph+1)*bin_h) (+ roi start), clipped to
    the feature map; likewise for columns;
  * the output is the maximum over the bin, or 0 for an empty bin.

PARITY UNPINNED: no fixture of the reference and no runnable MXNet pins this file.
"""
import numpy as np


def _round_away(x):
    return np.sign(x) * np.floor(np.abs(x) + np.float32(0.5))


def roi_pooling(data, rois, pooled_size=(7, 7), spatial_scale=0.0625):
    """data [B,C,H,W] float32, rois [R,5] (b,x1,y1,x2,y2) -> [R,C,ph,pw] float32."""
    data = np.asarray(data, dtype=np.float32)
    rois = np.asarray(rois, dtype=np.float32)
    B, C, H, W = data.shape
    PH, PW = pooled_size
    out = np.zeros((rois.shape[0], C, PH, PW), dtype=np.float32)
    s = np.float32(spatial_scale)
    for n, roi in enumerate(rois):
        b = int(roi[0])
        if b < 0 or b >= B:
            continue
        sw, sh, ew, eh = (int(_round_away(np.float32(v) * s)) for v in roi[1:5])
        rh, rw = max(eh - sh + 1, 1), max(ew - sw + 1, 1)
        bh = np.float32(rh) / np.float32(PH)
        bw = np.float32(rw) / np.float32(PW)
        for ph in range(PH):
            hs = int(np.floor(np.float32(ph) * bh)) + sh
            he = int(np.ceil(np.float32(ph + 1) * bh)) + sh
            hs, he = min(max(hs, 0), H), min(max(he, 0), H)
            for pw in range(PW):
                ws = int(np.floor(np.float32(pw) * bw)) + sw
                we = int(np.ceil(np.float32(pw + 1) * bw)) + sw
                ws, we = min(max(ws, 0), W), min(max(we, 0), W)
                if he > hs and we > ws:
                    out[n, :, ph, pw] = data[b, :, hs:he, ws:we].max(axis=(1, 2))
    return out
