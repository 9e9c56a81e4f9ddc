"""NumPy oracle for the box arithmetic of the detector (test infrastructure).

Restates, in independent code, what the reference computes in
  code/detection/symbol/processing.py:35-78    (delta decode, ``bbox_pred``)
  code/detection/symbol/processing.py:81-97    (``clip_boxes``)
  code/detection/symbol/processing.py:114-148  (greedy ``nms``)
  code/detection/symbol/processing.py:184-248  (``generate_anchors``)
  code/detection/symbol/proposal.py:28-142     (RPN proposal sequence)
  code/detection/detection.py:43-60            (detector post-processing)

Parity: PINNED for anchors / decode / clip / nms / the proposal sequence /
post-processing -- ``tests/golden/make_golden.py`` runs the reference's own
``processing`` module in the build container and the fixtures it wrote are
compared with this file in ``tests/test_oracle_golden.py``.

Two documented deviations, both where the reference is non-deterministic:
  * score ties: the reference sorts with an unstable ``argsort()[::-1]``
    (processing.py:129, proposal.py:106).  Here the order is descending score
    and, among equal scores, HIGHER original index first (what a stable
    ascending argsort gives once reversed).
  * padding: the reference pads a short keep list with ``numpy.random.choice``
    (proposal.py:118-120).  Here the pad is cyclic, ``keep[i % len(keep)]``.
"""
import numpy as np


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """[len(ratios)*len(scales), 4] float64 anchor table (processing.py:184-248).

    Ratio-major / scale-minor; each window keeps the centre of the
    (0, 0, base-1, base-1) box; sides are rounded half-to-even at the ratio step.
    """
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    side = float(base_size)
    ctr = 0.5 * (side - 1.0)
    area = side * side
    rows = []
    for r in ratios:
        w0 = np.round(np.sqrt(area / r))
        h0 = np.round(w0 * r)
        for s in scales:
            w, h = w0 * s, h0 * s
            rows.append([ctr - 0.5 * (w - 1), ctr - 0.5 * (h - 1),
                         ctr + 0.5 * (w - 1), ctr + 0.5 * (h - 1)])
    return np.asarray(rows, dtype=np.float64)


def bbox_pred(boxes, box_deltas, dtype=np.float64):
    """Decode [N,4] boxes with [N,4C] deltas -> [N,4C] (processing.py:35-78).

    The reference promotes the boxes to float64 (``astype(np.float)``) and
    writes into a float64 ``np.zeros``; the deltas stay float32, so ``exp`` is
    evaluated in float32 and only its result is widened (processing.py:65-66).
    ``dtype=np.float32`` gives the same steps entirely in single precision.
    """
    box_deltas = np.asarray(box_deltas)
    if boxes.shape[0] == 0:
        return np.zeros((0, box_deltas.shape[1]), dtype=dtype)
    b = np.asarray(boxes).astype(dtype)
    d = box_deltas.astype(dtype)
    e = box_deltas if box_deltas.dtype == np.float32 else d   # exp runs in the deltas' own dtype
    one, half = dtype(1.0), dtype(0.5)
    w = b[:, 2] - b[:, 0] + one
    h = b[:, 3] - b[:, 1] + one
    cx = b[:, 0] + half * (w - one)
    cy = b[:, 1] + half * (h - one)
    out = np.empty(d.shape, dtype=dtype)
    pcx = d[:, 0::4] * w[:, None] + cx[:, None]
    pcy = d[:, 1::4] * h[:, None] + cy[:, None]
    pw = np.exp(e[:, 2::4]).astype(dtype) * w[:, None]
    ph = np.exp(e[:, 3::4]).astype(dtype) * h[:, None]
    out[:, 0::4] = pcx - half * (pw - one)
    out[:, 1::4] = pcy - half * (ph - one)
    out[:, 2::4] = pcx + half * (pw - one)
    out[:, 3::4] = pcy + half * (ph - one)
    return out


def clip_boxes(boxes, im_shape):
    """Clamp x to [0, W-1] and y to [0, H-1] in place (processing.py:81-97)."""
    hmax = boxes.dtype.type(im_shape[0] - 1)
    wmax = boxes.dtype.type(im_shape[1] - 1)
    zero = boxes.dtype.type(0)
    for c, hi in ((0, wmax), (1, hmax), (2, wmax), (3, hmax)):
        boxes[:, c::4] = np.maximum(np.minimum(boxes[:, c::4], hi), zero)
    return boxes


def score_order(scores):
    """Descending score; equal scores -> higher original index first."""
    return np.argsort(np.asarray(scores), kind="stable")[::-1]


def nms(dets, thresh):
    """Greedy IoU suppression (processing.py:114-148) in ``dets.dtype``.

    Areas and overlaps use the +1 pixel-extent convention, the IoU is
    ``inter / (area_i + area_j - inter)`` and a box survives a kept box when
    ``iou <= thresh`` with ``thresh`` taken in the dtype of ``dets``.
    Returns indices into ``dets`` in the order they were kept.
    """
    dets = np.asarray(dets)
    n = dets.shape[0]
    if n == 0:
        return []
    t = dets.dtype.type
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    area = (x2 - x1 + t(1)) * (y2 - y1 + t(1))
    order = score_order(dets[:, 4])
    thr = t(thresh)
    alive = np.ones(n, dtype=bool)
    keep = []
    for pos in range(n):
        i = order[pos]
        if not alive[i]:
            continue
        keep.append(int(i))
        rest = order[pos + 1:]
        rest = rest[alive[rest]]
        if rest.size == 0:
            break
        iw = np.maximum(t(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + t(1))
        ih = np.maximum(t(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + t(1))
        inter = iw * ih
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[rest] - inter)
        alive[rest[~(iou <= thr)]] = False
    return keep


def shifted_anchors(height, width, feat_stride, anchors):
    """[(H*W*A), 4] anchors in (h, w, a) order (proposal.py:60-74)."""
    sx = np.arange(width, dtype=np.float64) * feat_stride
    sy = np.arange(height, dtype=np.float64) * feat_stride
    gx, gy = np.meshgrid(sx, sy)
    shifts = np.stack([gx.ravel(), gy.ravel(), gx.ravel(), gy.ravel()], axis=1)
    return (anchors[None, :, :] + shifts[:, None, :]).reshape(-1, 4)


def proposal(cls_prob, bbox_deltas, im_info, feat_stride=16, scales=(4, 8, 16, 32),
             ratios=(0.5, 1, 2), pre_nms_top_n=1000, post_nms_top_n=10,
             nms_thresh=0.7, min_size=16, dtype=np.float64, return_all=False):
    """RPN proposal sequence (proposal.py:28-130) for one image.

    cls_prob [1,2A,H,W] float32 (foreground = channels A:), bbox_deltas
    [1,4A,H,W] float32, im_info (h, w, scale).  Returns rois [post_nms_top_n,5]
    float32 (column 0 = batch index 0) and their scores [post_nms_top_n]
    float32.  With ``return_all`` also returns a dict with the intermediate
    lists (candidate order, kept indices) for stage-level checks.
    """
    cls_prob = np.asarray(cls_prob, dtype=np.float32)
    bbox_deltas = np.asarray(bbox_deltas, dtype=np.float32)
    im_info = np.asarray(im_info, dtype=np.float32).reshape(-1)[:3]
    if np.isnan(cls_prob).any():
        raise ValueError("there is nan in input scores")
    if np.isnan(bbox_deltas).any():
        raise ValueError("there is nan in input bbox_deltas")
    anchors0 = generate_anchors(float(feat_stride), ratios, scales)
    A = anchors0.shape[0]
    H, W = cls_prob.shape[-2:]
    scores = cls_prob[0, A:, :, :].transpose(1, 2, 0).reshape(-1)
    deltas = bbox_deltas[0, :, :H, :W].transpose(1, 2, 0).reshape(-1, 4)
    anchors = shifted_anchors(H, W, float(feat_stride), anchors0)

    boxes = bbox_pred(anchors, deltas, dtype=dtype)
    boxes = clip_boxes(boxes, (im_info[0], im_info[1]))
    lim = dtype(np.float32(min_size) * im_info[2])
    ws = boxes[:, 2] - boxes[:, 0] + dtype(1)
    hs = boxes[:, 3] - boxes[:, 1] + dtype(1)
    valid = np.where((ws >= lim) & (hs >= lim))[0]

    order = score_order(scores[valid])
    if pre_nms_top_n > 0:
        order = order[:pre_nms_top_n]
    cand = valid[order]                      # anchor index of each candidate, best first
    cboxes = boxes[cand]
    cscores = scores[cand]
    dets = np.concatenate([cboxes, cscores[:, None].astype(dtype)], axis=1)
    # candidates are already in tie-resolved order, so make that order explicit
    # for the greedy pass (a second sort of equal scores must not reshuffle them)
    keep = _nms_presorted(dets, nms_thresh)
    n_kept = len(keep)
    if post_nms_top_n > 0:
        keep = keep[:post_nms_top_n]
        if 0 < len(keep) < post_nms_top_n:
            keep = [keep[i % len(keep)] for i in range(post_nms_top_n)]
    n_out = post_nms_top_n if post_nms_top_n > 0 else len(keep)
    rois = np.zeros((n_out, 5), dtype=np.float32)
    out_scores = np.zeros((n_out,), dtype=np.float32)
    if keep:
        rois[:, 1:] = cboxes[keep].astype(np.float32)
        out_scores[:] = cscores[keep]
    if return_all:
        return rois, out_scores, dict(candidates=cand, keep=np.asarray(keep, dtype=np.int64),
                                      n_valid=valid.size, n_kept=n_kept, boxes=boxes)
    return rois, out_scores


def _nms_presorted(dets, thresh):
    """``nms`` for rows already in the order the greedy pass must visit them."""
    n = dets.shape[0]
    if n == 0:
        return []
    # strictly decreasing surrogate scores reproduce the given order in nms()
    tmp = dets.copy()
    tmp[:, 4] = np.arange(n, 0, -1, dtype=dets.dtype)
    return nms(tmp, thresh)


def postprocess(rois, scores, bbox_deltas, im_shape, thresh=0.8, nms_thresh=0.3, do_round=True):
    """Detector tail (detection.py:43-60).

    rois [R,4] float32 (x1,y1,x2,y2), scores [R,2] float32, bbox_deltas [R,8]
    float32, im_shape (H, W).  float64 decode and clip, class-1 columns, keep
    ``score >= float32(thresh)``, cast to float32, greedy NMS in float32, rows
    in kept order, coordinates rounded half-to-even.  Returns float32 [n,5].
    """
    rois = np.asarray(rois, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    bbox_deltas = np.asarray(bbox_deltas, dtype=np.float32)
    pred = bbox_pred(rois, bbox_deltas, dtype=np.float64)
    pred = clip_boxes(pred, (float(im_shape[0]), float(im_shape[1])))
    cls_boxes = pred[:, 4:8]
    cls_scores = scores[:, 1]
    sel = np.where(cls_scores >= np.float32(thresh))[0]
    dets = np.concatenate([cls_boxes[sel], cls_scores[sel, None].astype(np.float64)], axis=1).astype(np.float32)
    keep = nms(dets, nms_thresh)
    dets = dets[keep, :]
    if do_round:
        dets[:, :4] = np.round(dets[:, :4]).astype(np.int32)
    return dets
