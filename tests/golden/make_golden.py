#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's own code.

Run in the build container only (it needs /root/reference):

    python tests/golden/make_golden.py

It imports the reference's NumPy-only box module
(code/detection/symbol/processing.py) untouched, calls its functions on seeded
inputs and stores inputs + outputs as small .npz files.  The module was written
for Python 2 / old NumPy, so two names it expects are provided from here before
the calls: ``numpy.float`` (-> float) and the builtin ``xrange`` (-> range).

The RPN proposal sequence lives in code/detection/symbol/proposal.py, which
cannot be imported (it needs mxnet).  Its forward pass (proposal.py:28-130) is
therefore driven step by step from here with the reference's own bbox_pred /
clip_boxes / clip_pad / nms / generate_anchors doing every computation; only
the glue (transposes, the min-size filter of proposal.py:136-142, top-N cuts)
is written out.  Random padding (proposal.py:118-120) is not reproduced: the
fixtures store the un-padded keep list.

All score vectors are tie-free by construction (they are permutations), so the
reference's unstable sort cannot make the expected outputs ambiguous; one
extra fixture has deliberate ties and is only ever compared as a set.
"""
import builtins
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/code")
np.float = float            # name removed from NumPy >= 1.24
builtins.xrange = range     # Python 2 builtin
from detection.symbol import processing as ref  # noqa: E402


def unique_scores(rng, n, lo=0.0, hi=1.0):
    """n distinct float32 values in (lo, hi), shuffled."""
    s = (rng.permutation(n).astype(np.float64) + 0.5) / n
    s = (lo + (hi - lo) * s).astype(np.float32)
    assert np.unique(s).size == n
    return s


def clustered_dets(rng, n, n_centres, width=1920, height=1080, size_lo=20.0, size_hi=200.0):
    """float32 [n,5] boxes jittered around a few centres so that many overlap."""
    cx = rng.uniform(0, width, n_centres)
    cy = rng.uniform(0, height, n_centres)
    sz = rng.uniform(size_lo, size_hi, n_centres)
    which = rng.integers(0, n_centres, n)
    s = sz[which] * rng.uniform(0.8, 1.25, n)
    asp = rng.uniform(0.7, 1.4, n)
    x = cx[which] + rng.normal(0, 0.15, n) * s
    y = cy[which] + rng.normal(0, 0.15, n) * s
    w, h = s * np.sqrt(asp), s / np.sqrt(asp)
    dets = np.stack([x - w / 2, y - h / 2, x + w / 2, y + h / 2, np.zeros(n)], axis=1).astype(np.float32)
    dets[:, 4] = unique_scores(rng, n)
    return dets


def gold_anchors(out):
    out["anchors_16_r3_s4"] = ref.generate_anchors(base_size=16.0, ratios=[0.5, 1, 2],
                                                   scales=np.array([4.0, 8.0, 16.0, 32.0]))
    out["anchors_16_r3_s3"] = ref.generate_anchors(base_size=16, ratios=[0.5, 1, 2],
                                                   scales=2 ** np.arange(3, 6))


def gold_decode(out):
    rng = np.random.default_rng(11)
    n = 512
    x1 = rng.uniform(-60, 680, n)
    y1 = rng.uniform(-60, 500, n)
    boxes = np.stack([x1, y1, x1 + rng.uniform(4, 300, n), y1 + rng.uniform(4, 300, n)], axis=1).astype(np.float32)
    for ncls, tag in ((1, "c1"), (2, "c2")):
        deltas = rng.normal(0, 0.5, (n, 4 * ncls)).astype(np.float32)
        deltas[::7, 2::4] *= 4.0         # a few large log-size deltas
        pred = ref.bbox_pred(boxes.copy(), deltas.copy())
        clipped = ref.clip_boxes(pred.copy(), (480, 640))
        out["decode_%s_boxes" % tag] = boxes
        out["decode_%s_deltas" % tag] = deltas
        out["decode_%s_pred" % tag] = pred
        out["decode_%s_clipped" % tag] = clipped
    empty = ref.bbox_pred(np.zeros((0, 4), np.float32), np.zeros((0, 8), np.float32))
    out["decode_empty_shape"] = np.asarray(empty.shape)


def gold_nms(out):
    cases = [(0, 1), (1, 1), (10, 3), (1000, 40), (6000, 200)]
    for n, nc in cases:
        rng = np.random.default_rng(100 + n)
        dets = clustered_dets(rng, n, nc) if n else np.zeros((0, 5), np.float32)
        out["nms_n%d_dets" % n] = dets
        for thr in (0.3, 0.7):
            keep = ref.nms(dets.astype(np.float32), thr)
            out["nms_n%d_t%d_keep" % (n, int(thr * 10))] = np.asarray(keep, dtype=np.int64)
    # float64 input (what the proposal twin feeds nms through np.hstack)
    rng = np.random.default_rng(77)
    d64 = clustered_dets(rng, 1500, 60).astype(np.float64)
    d64[:, :4] += rng.uniform(0, 1e-3, (1500, 4))
    out["nms64_dets"] = d64
    out["nms64_t7_keep"] = np.asarray(ref.nms(d64, 0.7), dtype=np.int64)
    # deliberate ties: duplicates of rows with identical scores -> only the SET of
    # kept boxes (coordinates) is defined, not which duplicate index is reported
    rng = np.random.default_rng(5)
    base = clustered_dets(rng, 40, 6)
    tied = np.concatenate([base, base[:15], base[:5]], axis=0)
    out["nms_tied_dets"] = tied
    out["nms_tied_t3_keep"] = np.asarray(ref.nms(tied, 0.3), dtype=np.int64)


def run_twin(cls_prob, bbox_deltas, im_info, pre_n, post_n, nms_thresh, min_size,
             feat_stride=16.0, scales=(4.0, 8.0, 16.0, 32.0), ratios=(0.5, 1, 2)):
    """proposal.py:28-130 driven from here; every computation is a reference call."""
    anchors0 = ref.generate_anchors(base_size=feat_stride, scales=np.array(scales), ratios=list(ratios))
    A = anchors0.shape[0]
    scores = cls_prob[:, A:, :, :]
    height, width = scores.shape[-2:]
    sx = np.arange(0, width) * feat_stride
    sy = np.arange(0, height) * feat_stride
    sx, sy = np.meshgrid(sx, sy)
    shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
    K = shifts.shape[0]
    anchors = (anchors0.reshape((1, A, 4)) + shifts.reshape((1, K, 4)).transpose((1, 0, 2))).reshape((K * A, 4))
    deltas = ref.clip_pad(bbox_deltas, (height, width)).transpose((0, 2, 3, 1)).reshape((-1, 4))
    scores = scores.transpose((0, 2, 3, 1)).reshape((-1, 1))
    props = ref.bbox_pred(anchors, deltas)
    props = ref.clip_boxes(props, im_info[:2])
    lim = min_size * im_info[2]
    ws = props[:, 2] - props[:, 0] + 1
    hs = props[:, 3] - props[:, 1] + 1
    valid = np.where((ws >= lim) & (hs >= lim))[0]
    props_v, scores_v = props[valid, :], scores[valid]
    order = scores_v.ravel().argsort()[::-1]
    if pre_n > 0:
        order = order[:pre_n]
    props_o, scores_o = props_v[order, :], scores_v[order]
    keep = ref.nms(np.hstack((props_o, scores_o)), nms_thresh)
    n_kept = len(keep)
    if post_n > 0:
        keep = keep[:post_n]
    rois = props_o[keep, :].astype(np.float32)
    return dict(rois=rois, scores=scores_o[keep].ravel().astype(np.float32),
                cand=valid[order].astype(np.int64), keep=np.asarray(keep, dtype=np.int64),
                n_valid=np.int64(valid.size), n_kept=np.int64(n_kept), all_boxes=props)


def gold_proposal(out):
    grids = {"g8x12": (8, 12, (128.0, 192.0, 1.0)), "g38x50": (38, 50, (600.0, 800.0, 1.25))}
    for gname, (H, W, info) in grids.items():
        rng = np.random.default_rng(1000 + H)
        A = 12
        fg = unique_scores(rng, A * H * W).reshape(A, H, W)
        cls_prob = np.concatenate([1.0 - fg, fg], axis=0)[None].astype(np.float32)
        bbox = rng.normal(0, 0.35, (1, 4 * A, H, W)).astype(np.float32)
        im_info = np.asarray(info, dtype=np.float32)
        out["prop_%s_cls_prob" % gname] = cls_prob
        out["prop_%s_bbox" % gname] = bbox
        out["prop_%s_im_info" % gname] = im_info
        for min_size in (16, 24, 50):
            for pre_n, post_n in ((1000, 10), (6000, 300)):
                r = run_twin(cls_prob, bbox, im_info, pre_n, post_n, 0.7, min_size)
                tag = "prop_%s_m%d_p%d_q%d" % (gname, min_size, pre_n, post_n)
                out[tag + "_rois"] = r["rois"]
                out[tag + "_scores"] = r["scores"]
                out[tag + "_keep"] = r["keep"]
                out[tag + "_cand_head"] = r["cand"][:64]
                out[tag + "_counts"] = np.asarray([r["n_valid"], r["cand"].size, r["n_kept"]], dtype=np.int64)
        # decoded+clipped boxes of every anchor for the small grid (checks (h,w,a) order)
        if gname == "g8x12":
            out["prop_g8x12_all_boxes"] = run_twin(cls_prob, bbox, im_info, 1000, 10, 0.7, 16)["all_boxes"]


def gold_postprocess(out):
    """detection.py:43-60 with the reference's functions on seeded head outputs."""
    for R, seed in ((10, 3), (300, 4)):
        rng = np.random.default_rng(seed)
        Hh, Ww = 480, 640
        x1 = rng.uniform(0, Ww - 80, R)
        y1 = rng.uniform(0, Hh - 80, R)
        rois = np.stack([x1, y1, x1 + rng.uniform(30, 200, R), y1 + rng.uniform(30, 200, R)], axis=1)
        rois = ref.clip_boxes(rois, (Hh, Ww)).astype(np.float32)
        if R > 10:
            rois[R // 2:] = rois[:R - R // 2] + rng.uniform(-6, 6, (R - R // 2, 4)).astype(np.float32)
        fg = unique_scores(rng, R, 0.5, 1.0)
        scores = np.stack([1 - fg, fg], axis=1).astype(np.float32)
        deltas = rng.normal(0, 0.2, (R, 8)).astype(np.float32)
        pred = ref.bbox_pred(rois, deltas)
        pred = ref.clip_boxes(pred, (Hh, Ww))
        cls_boxes = pred[:, 4:8]
        cls_scores = scores[:, 1]
        sel = np.where(cls_scores >= 0.8)[0]
        dets = np.hstack((cls_boxes[sel, :], cls_scores[sel][:, np.newaxis])).astype(np.float32)
        keep = ref.nms(dets.astype(np.float32), 0.3)
        dets = dets[keep, :]
        unrounded = dets.copy()
        dets[:, :4] = (dets[:, :4]).round().astype(np.int32)
        tag = "post_r%d" % R
        out[tag + "_rois"] = rois
        out[tag + "_scores"] = scores
        out[tag + "_deltas"] = deltas
        out[tag + "_im_shape"] = np.asarray([Hh, Ww], dtype=np.int64)
        out[tag + "_dets"] = dets
        out[tag + "_dets_unrounded"] = unrounded


def main():
    groups = {"anchors": gold_anchors, "decode": gold_decode, "nms": gold_nms,
              "proposal": gold_proposal, "postprocess": gold_postprocess}
    for name, fn in groups.items():
        out = {}
        fn(out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-12s %3d arrays  %7.1f KiB" % (name, len(out), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
