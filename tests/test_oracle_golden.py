"""The NumPy oracle must reproduce every vector the reference's own box module
produced (tests/golden/*.npz, written by tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import boxes as ob


def test_anchor_tables(golden):
    g = golden("anchors")
    a = ob.generate_anchors(16.0, (0.5, 1, 2), (4.0, 8.0, 16.0, 32.0))
    assert a.shape == (12, 4)
    np.testing.assert_array_equal(a, g["anchors_16_r3_s4"])
    np.testing.assert_array_equal(a[0], [-38, -16, 53, 31])
    np.testing.assert_array_equal(a[-1], [-168, -344, 183, 359])
    np.testing.assert_array_equal(ob.generate_anchors(16, (0.5, 1, 2), (8, 16, 32)), g["anchors_16_r3_s3"])


@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_decode_and_clip_bit_exact(golden, tag):
    g = golden("decode")
    pred = ob.bbox_pred(g["decode_%s_boxes" % tag], g["decode_%s_deltas" % tag])
    assert pred.dtype == np.float64
    np.testing.assert_array_equal(pred, g["decode_%s_pred" % tag])
    clipped = ob.clip_boxes(pred.copy(), (480, 640))
    np.testing.assert_array_equal(clipped, g["decode_%s_clipped" % tag])


def test_decode_empty(golden):
    g = golden("decode")
    out = ob.bbox_pred(np.zeros((0, 4), np.float32), np.zeros((0, 8), np.float32))
    assert tuple(out.shape) == tuple(g["decode_empty_shape"])


@pytest.mark.parametrize("n", [0, 1, 10, 1000, 6000])
@pytest.mark.parametrize("thr", [0.3, 0.7])
def test_nms_keep_lists(golden, n, thr):
    g = golden("nms")
    dets = g["nms_n%d_dets" % n]
    keep = ob.nms(dets, thr)
    np.testing.assert_array_equal(np.asarray(keep, dtype=np.int64), g["nms_n%d_t%d_keep" % (n, int(thr * 10))])


def test_nms_float64(golden):
    g = golden("nms")
    keep = ob.nms(g["nms64_dets"], 0.7)
    np.testing.assert_array_equal(np.asarray(keep, dtype=np.int64), g["nms64_t7_keep"])


def test_nms_ties_same_box_set(golden):
    g = golden("nms")
    dets = g["nms_tied_dets"]
    mine = dets[ob.nms(dets, 0.3)]
    theirs = dets[g["nms_tied_t3_keep"]]
    assert sorted(map(tuple, mine)) == sorted(map(tuple, theirs))


@pytest.mark.parametrize("grid", ["g8x12", "g38x50"])
@pytest.mark.parametrize("min_size", [16, 24, 50])
@pytest.mark.parametrize("pre_post", [(1000, 10), (6000, 300)])
def test_proposal_sequence(golden, grid, min_size, pre_post):
    g = golden("proposal")
    pre_n, post_n = pre_post
    tag = "prop_%s_m%d_p%d_q%d" % (grid, min_size, pre_n, post_n)
    rois, scores, info = ob.proposal(g["prop_%s_cls_prob" % grid], g["prop_%s_bbox" % grid],
                                     g["prop_%s_im_info" % grid], pre_nms_top_n=pre_n, post_nms_top_n=post_n,
                                     nms_thresh=0.7, min_size=min_size, return_all=True)
    n_valid, n_cand, n_kept = g[tag + "_counts"]
    assert info["n_valid"] == n_valid and info["candidates"].size == n_cand and info["n_kept"] == n_kept
    np.testing.assert_array_equal(info["candidates"][:64], g[tag + "_cand_head"])
    m = min(n_kept, post_n)
    np.testing.assert_array_equal(info["keep"][:m], g[tag + "_keep"][:m])
    assert rois.shape == (post_n, 5) and rois.dtype == np.float32
    np.testing.assert_array_equal(rois[:m, 1:], g[tag + "_rois"][:m])
    np.testing.assert_array_equal(scores[:m], g[tag + "_scores"][:m])
    assert (rois[:, 0] == 0).all()
    if 0 < m < post_n:          # cyclic padding of a short list
        np.testing.assert_array_equal(rois[m:], rois[np.arange(m, post_n) % m])


def test_proposal_anchor_order(golden):
    g = golden("proposal")
    _, _, info = ob.proposal(g["prop_g8x12_cls_prob"], g["prop_g8x12_bbox"], g["prop_g8x12_im_info"],
                             return_all=True)
    np.testing.assert_array_equal(info["boxes"], g["prop_g8x12_all_boxes"])


def test_proposal_rejects_nan(golden):
    g = golden("proposal")
    bad = g["prop_g8x12_cls_prob"].copy()
    bad[0, 13, 2, 3] = np.nan
    with pytest.raises(ValueError):
        ob.proposal(bad, g["prop_g8x12_bbox"], g["prop_g8x12_im_info"])


@pytest.mark.parametrize("R", [10, 300])
def test_postprocess(golden, R):
    g = golden("postprocess")
    tag = "post_r%d" % R
    args = (g[tag + "_rois"], g[tag + "_scores"], g[tag + "_deltas"], tuple(g[tag + "_im_shape"]))
    dets = ob.postprocess(*args)
    assert dets.dtype == np.float32
    np.testing.assert_array_equal(dets, g[tag + "_dets"])
    np.testing.assert_array_equal(ob.postprocess(*args, do_round=False), g[tag + "_dets_unrounded"])


@pytest.mark.parametrize("n", [0, 1, 10, 1000, 6000])
@pytest.mark.parametrize("thr", [0.3, 0.7])
def test_c_nms_matches_reference_vectors(golden, n, thr):
    from oracle import cboxes
    g = golden("nms")
    keep = cboxes.nms(g["nms_n%d_dets" % n], thr)
    np.testing.assert_array_equal(np.asarray(keep, dtype=np.int64), g["nms_n%d_t%d_keep" % (n, int(thr * 10))])


def test_c_nms_float64_and_ties(golden):
    from oracle import cboxes
    g = golden("nms")
    np.testing.assert_array_equal(cboxes.nms(g["nms64_dets"], 0.7), g["nms64_t7_keep"])
    np.testing.assert_array_equal(cboxes.nms(g["nms_tied_dets"], 0.3), ob.nms(g["nms_tied_dets"], 0.3))
