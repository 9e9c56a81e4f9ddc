"""GPU parity of the box stages, called through the C ABI, against (a) the golden vectors
the reference's own module produced and (b) the NumPy oracle on fresh seeded inputs."""
import numpy as np
import pytest

from oracle import boxes as ob
from oracle import roipool as orp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    from facedetection_amd.symbol import processing
    return processing


@pytest.fixture(scope="module")
def ops():
    from facedetection_amd import ops
    return ops


def rand_dets(rng, n, n_centres, dtype=np.float32, w=1920, h=1080, lo=20.0, hi=200.0):
    cx, cy = rng.uniform(0, w, n_centres), rng.uniform(0, h, n_centres)
    sz = rng.uniform(lo, hi, n_centres)
    k = rng.integers(0, n_centres, n)
    s = sz[k] * rng.uniform(0.8, 1.25, n)
    x, y = cx[k] + rng.normal(0, 0.15, n) * s, cy[k] + rng.normal(0, 0.15, n) * s
    d = np.stack([x - s / 2, y - s / 2, x + s / 2, y + s / 2, np.zeros(n)], axis=1).astype(dtype)
    d[:, 4] = ((rng.permutation(n) + 0.5) / n).astype(dtype)
    return d


# ---------------------------------------------------------------- decode / clip
@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_bbox_pred_and_clip_vs_reference_vectors(P, golden, tag):
    g = golden("decode")
    pred = P.bbox_pred(g["decode_%s_boxes" % tag], g["decode_%s_deltas" % tag])
    assert pred.dtype == np.float64
    ref = g["decode_%s_pred" % tag]
    # exp() is float32 on both sides but from different libraries (1-2 ulp apart) and that
    # relative error is multiplied by the predicted box size: allow 3 float32 ulps of it
    size = np.maximum(ref[:, 2::4] - ref[:, 0::4], ref[:, 3::4] - ref[:, 1::4]) + 1.0
    tol = np.repeat(3 * 2.0 ** -23 * size, 4, axis=1) + 1e-9
    assert (np.abs(pred - ref) <= tol).all()
    small = size < 2000.0                       # for ordinary boxes that is below 1e-3 px
    assert np.abs(pred - ref)[np.repeat(small, 4, axis=1)].max() < 1e-3
    # the clamp itself is exact: clip the reference's decoded boxes on the GPU
    clipped = P.clip_boxes(ref.copy(), (480, 640))
    np.testing.assert_array_equal(clipped, g["decode_%s_clipped" % tag])


def test_bbox_pred_zero_deltas_is_exact(P):
    rng = np.random.default_rng(2)
    x1, y1 = rng.integers(0, 500, 300), rng.integers(0, 400, 300)
    boxes = np.stack([x1, y1, x1 + rng.integers(1, 200, 300), y1 + rng.integers(1, 200, 300)], 1).astype(np.float32)
    out = P.bbox_pred(boxes, np.zeros((300, 8), np.float32))
    np.testing.assert_array_equal(out, ob.bbox_pred(boxes, np.zeros((300, 8), np.float32)))
    np.testing.assert_array_equal(out[:, :4], boxes.astype(np.float64))


def test_bbox_pred_empty(P):
    assert P.bbox_pred(np.zeros((0, 4), np.float32), np.zeros((0, 8), np.float32)).shape == (0, 8)


# ---------------------------------------------------------------- nms
@pytest.mark.parametrize("n", [0, 1, 10, 1000, 6000])
@pytest.mark.parametrize("thr", [0.3, 0.7])
def test_nms_keep_lists_vs_reference_vectors(P, golden, n, thr):
    g = golden("nms")
    keep = P.nms(g["nms_n%d_dets" % n], thr)
    np.testing.assert_array_equal(np.asarray(keep, dtype=np.int64), g["nms_n%d_t%d_keep" % (n, int(thr * 10))])


def test_nms_float64_vs_reference_vectors(P, golden):
    g = golden("nms")
    keep = P.nms(g["nms64_dets"], 0.7)
    np.testing.assert_array_equal(np.asarray(keep, dtype=np.int64), g["nms64_t7_keep"])


def test_nms_ties_same_box_set_as_reference(P, golden):
    g = golden("nms")
    dets = g["nms_tied_dets"]
    mine = dets[P.nms(dets, 0.3)]
    theirs = dets[g["nms_tied_t3_keep"]]
    assert sorted(map(tuple, mine)) == sorted(map(tuple, theirs))
    # and index-exact against the oracle, which shares the tie rule
    np.testing.assert_array_equal(P.nms(dets, 0.3), ob.nms(dets, 0.3))


@pytest.mark.parametrize("n,nc,dtype", [(63, 5, np.float32), (64, 5, np.float32), (65, 5, np.float32),
                                        (2049, 90, np.float32), (4097, 50, np.float64), (20000, 1500, np.float32)])
def test_nms_vs_oracle_seeded(P, n, nc, dtype):
    rng = np.random.default_rng(n)
    dets = rand_dets(rng, n, nc, dtype)
    for thr in (0.3, 0.7):
        np.testing.assert_array_equal(P.nms(dets, thr), ob.nms(dets, thr))


@pytest.mark.parametrize("n,nc", [(100000, 1000), (300000, 300)])
def test_nms_many_row_panels_vs_c_oracle(P, n, nc):
    """Tens of row panels, the `removed` bitmap carried between them; index-exact against the
    C restatement of the greedy loop (the NumPy one needs minutes at this size)."""
    from oracle import cboxes
    rng = np.random.default_rng(n + 1)
    dets = rand_dets(rng, n, nc, lo=12.0, hi=40.0)
    np.testing.assert_array_equal(np.asarray(P.nms(dets, 0.7), dtype=np.int64),
                                  np.asarray(cboxes.nms(dets, 0.7), dtype=np.int64))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("thr", [0.5, 0.3, 0.25, 0.7, 0.2])
def test_nms_pairs_exactly_on_the_threshold(P, dtype, thr):
    """Boxes on a small integer grid: IoUs are ratios of small integers, so many pairs sit exactly
    on the threshold (1/2, 3/10, 1/4, 7/10, 1/5) or one unit of area away from it.  The kernel decides
    pairs far from the threshold without the division; these are the pairs that must still take the
    reference's rounded quotient, `inter / union <= thresh` in the dtype of the boxes."""
    rng = np.random.default_rng(int(thr * 1000) + (dtype == np.float64))
    n = 6000                                      # two row panels
    x1 = rng.integers(0, 24, n); y1 = rng.integers(0, 24, n)
    w = rng.integers(1, 21, n); h = rng.integers(1, 21, n)
    dets = np.stack([x1, y1, x1 + w - 1, y1 + h - 1, rng.permutation(n) / n], axis=1).astype(dtype)
    # how many pairs of the first 1000 boxes sit exactly on the threshold (the test must not be vacuous)
    a = dets[:1000].astype(np.float64)
    area = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    iw = np.clip(np.minimum(a[:, None, 2], a[None, :, 2]) - np.maximum(a[:, None, 0], a[None, :, 0]) + 1, 0, None)
    ih = np.clip(np.minimum(a[:, None, 3], a[None, :, 3]) - np.maximum(a[:, None, 1], a[None, :, 1]) + 1, 0, None)
    inter = iw * ih
    union = area[:, None] + area[None, :] - inter
    from fractions import Fraction
    fr = Fraction(thr).limit_denominator(20)
    assert int(np.sum(inter * fr.denominator == union * fr.numerator)) >= 20
    from oracle import cboxes
    want = cboxes.nms(dets, thr)
    if thr == 0.5:
        assert want == list(ob.nms(dets, thr))    # the C and the NumPy restatement agree on these as well
    np.testing.assert_array_equal(np.asarray(P.nms(dets, thr), dtype=np.int64), np.asarray(want, dtype=np.int64))


def test_nms_many_equal_scores(P):
    rng = np.random.default_rng(9)
    dets = rand_dets(rng, 3000, 100)
    dets[:, 4] = np.float32(1.0)          # saturated scores: order is by index alone
    dets[::3, 4] = np.float32(0.5)
    np.testing.assert_array_equal(P.nms(dets, 0.5), ob.nms(dets, 0.5))


def test_nms_properties_large(P):
    """Size-independent checks at a size the oracle does not run: kept boxes are pairwise
    below the threshold, come in descending score order, and the call is idempotent."""
    rng = np.random.default_rng(123)
    n = 100000
    dets = rand_dets(rng, n, 5000, lo=12.0, hi=40.0)
    keep = np.asarray(P.nms(dets, 0.7))
    assert 0 < keep.size <= n and np.unique(keep).size == keep.size
    assert (np.diff(dets[keep, 4]) <= 0).all()
    again = np.asarray(P.nms(dets[keep], 0.7))
    np.testing.assert_array_equal(again, np.arange(keep.size))
    # a suppressed box overlaps some better kept box above the threshold (sample)
    gone = np.setdiff1d(np.arange(n), keep)[:200]
    kb = dets[keep]
    t = np.float32
    for j in gone:
        better = kb[kb[:, 4] >= dets[j, 4]]
        iw = np.maximum(t(0), np.minimum(better[:, 2], dets[j, 2]) - np.maximum(better[:, 0], dets[j, 0]) + t(1))
        ih = np.maximum(t(0), np.minimum(better[:, 3], dets[j, 3]) - np.maximum(better[:, 1], dets[j, 1]) + t(1))
        a = (better[:, 2] - better[:, 0] + t(1)) * (better[:, 3] - better[:, 1] + t(1))
        aj = (dets[j, 2] - dets[j, 0] + t(1)) * (dets[j, 3] - dets[j, 1] + t(1))
        iou = iw * ih / (a + aj - iw * ih)
        assert (iou > t(0.7)).any()


# ---------------------------------------------------------------- proposal
@pytest.mark.parametrize("grid", ["g8x12", "g38x50"])
@pytest.mark.parametrize("min_size", [16, 24, 50])
@pytest.mark.parametrize("pre_post", [(1000, 10), (6000, 300)])
def test_proposal_vs_reference_vectors(ops, golden, grid, min_size, pre_post):
    g = golden("proposal")
    pre_n, post_n = pre_post
    tag = "prop_%s_m%d_p%d_q%d" % (grid, min_size, pre_n, post_n)
    rois, scores, counts = ops.proposal(g["prop_%s_cls_prob" % grid], g["prop_%s_bbox" % grid],
                                        g["prop_%s_im_info" % grid], output_score=True, return_counts=True,
                                        rpn_pre_nms_top_n=pre_n, rpn_post_nms_top_n=post_n, threshold=0.7,
                                        rpn_min_size=min_size)
    n_valid, n_cand, n_kept = g[tag + "_counts"]
    m = min(n_kept, post_n)
    assert tuple(counts[0]) == (n_valid, n_cand, m)
    assert rois.shape == (post_n, 5) and (rois[:, 0] == 0).all()
    # same boxes in the same order: scores identify them exactly, coordinates agree to float32 exp() rounding
    np.testing.assert_array_equal(scores[:m, 0], g[tag + "_scores"][:m])
    np.testing.assert_allclose(rois[:m, 1:], g[tag + "_rois"][:m], rtol=1e-6, atol=1e-3)
    if 0 < m < post_n:
        np.testing.assert_array_equal(rois[m:], rois[np.arange(m, post_n) % m])


@pytest.mark.parametrize("f64", [True, False])
def test_proposal_vs_oracle_batch_and_ties(ops, f64):
    rng = np.random.default_rng(31)
    B, A, H, W = 3, 12, 30, 40
    logits = rng.normal(0, 3.0, (B, 2 * A, H, W)).astype(np.float32)
    e = np.exp(logits - logits.max())
    fg = e[:, A:] / (e[:, :A] + e[:, A:])
    fg = np.round(fg * 64) / 64               # coarse scores: thousands of exact ties
    cls_prob = np.concatenate([1 - fg, fg], axis=1).astype(np.float32)
    bbox = rng.normal(0, 0.3, (B, 4 * A, H, W)).astype(np.float32)
    info = np.array([[480, 640, 1.0]] * B, dtype=np.float32)
    rois, scores, counts = ops.proposal(cls_prob, bbox, info, output_score=True, return_counts=True,
                                        rpn_pre_nms_top_n=2000, rpn_post_nms_top_n=100, threshold=0.7,
                                        rpn_min_size=16, box_f64=f64)
    dt = np.float64 if f64 else np.float32
    for b in range(B):
        r, s, inf = ob.proposal(cls_prob[b:b + 1], bbox[b:b + 1], info[b], pre_nms_top_n=2000, post_nms_top_n=100,
                                nms_thresh=0.7, min_size=16, dtype=dt, return_all=True)
        assert counts[b, 0] == inf["n_valid"] and counts[b, 1] == inf["candidates"].size
        blk = slice(b * 100, (b + 1) * 100)
        assert (rois[blk, 0] == b).all()
        np.testing.assert_array_equal(scores[blk, 0], s)
        np.testing.assert_allclose(rois[blk, 1:], r[:, 1:], rtol=1e-6, atol=1e-3)


def test_proposal_no_valid_box_gives_zero_rois(ops):
    A, H, W = 12, 6, 7
    cls_prob = np.full((1, 2 * A, H, W), 0.5, np.float32)
    bbox = np.zeros((1, 4 * A, H, W), np.float32)
    rois, counts = ops.proposal(cls_prob, bbox, np.array([[96, 112, 1.0]], np.float32), return_counts=True,
                                rpn_min_size=5000)
    assert tuple(counts[0]) == (0, 0, 0) and not rois.any()


def test_proposal_rejects_nan_like_the_reference(ops, golden):
    g = golden("proposal")
    bad = g["prop_g8x12_bbox"].copy()
    bad[0, 5, 1, 1] = np.nan
    with pytest.raises(ValueError):
        ops.proposal(g["prop_g8x12_cls_prob"], bad, g["prop_g8x12_im_info"])


def test_proposal_operator_protocol(golden):
    from facedetection_amd.symbol.config import config
    from facedetection_amd.symbol.proposal import ProposalProp
    g = golden("proposal")
    old = config.TEST.RPN_MIN_SIZE
    config.TEST.RPN_MIN_SIZE = 24
    try:
        op = ProposalProp("16", "(4, 8, 16, 32)", "(0.5, 1, 2)", output_score=True).create_operator()
        out = []
        op.forward(False, ["write", "write"], [g["prop_g38x50_cls_prob"], g["prop_g38x50_bbox"],
                                               g["prop_g38x50_im_info"][None]], out)
    finally:
        config.TEST.RPN_MIN_SIZE = old
    tag = "prop_g38x50_m24_p1000_q10"
    np.testing.assert_array_equal(out[1][:, 0], g[tag + "_scores"])
    np.testing.assert_allclose(out[0][:, 1:], g[tag + "_rois"], rtol=1e-6, atol=1e-3)


# ---------------------------------------------------------------- roi pooling
def test_roipool_vs_oracle(ops):
    rng = np.random.default_rng(8)
    B, C, H, W = 2, 32, 30, 40
    data = rng.normal(0, 1, (B, C, H, W)).astype(np.float32)
    R = 40
    x1, y1 = rng.uniform(-20, 600, R), rng.uniform(-20, 440, R)
    rois = np.stack([rng.integers(0, B, R), x1, y1, x1 + rng.uniform(0, 300, R), y1 + rng.uniform(0, 300, R)],
                    axis=1).astype(np.float32)
    rois[0, 1:] = [8, 8, 8, 8]                 # single-cell roi
    rois[1, 1:] = [700, 500, 800, 600]         # fully outside -> empty bins -> 0
    rois[2, 1:] = [0, 0, 639, 479]             # whole map
    out = ops.roi_pooling(data, rois, (7, 7), 0.0625)
    np.testing.assert_array_equal(out, orp.roi_pooling(data, rois, (7, 7), 0.0625))


# ---------------------------------------------------------------- detector tail
@pytest.mark.parametrize("R", [10, 300])
def test_postprocess_vs_reference_vectors(ops, golden, R):
    g = golden("postprocess")
    tag = "post_r%d" % R
    args = (g[tag + "_rois"], g[tag + "_scores"], g[tag + "_deltas"], tuple(g[tag + "_im_shape"]))
    dets = ops.postprocess(*args)
    ref = g[tag + "_dets"]
    assert dets.dtype == np.float32 and dets.shape == ref.shape
    np.testing.assert_array_equal(dets[:, 4], ref[:, 4])          # same boxes kept, same order
    assert np.abs(dets[:, :4] - ref[:, :4]).max() <= 1.0          # rounding of a coordinate 1 ulp apart
    assert (dets[:, :4] == ref[:, :4]).mean() > 0.99
    raw = ops.postprocess(*args, do_round=False)
    np.testing.assert_allclose(raw, g[tag + "_dets_unrounded"], rtol=1e-6, atol=1e-3)


def test_postprocess_nothing_above_threshold(ops):
    rois = np.array([[10, 10, 50, 50]], np.float32)
    dets = ops.postprocess(rois, np.array([[0.9, 0.1]], np.float32), np.zeros((1, 8), np.float32), (100, 100))
    assert dets.shape == (0, 5)
