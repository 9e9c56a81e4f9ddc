"""The SORT-compatible consumer of the detection lists (host code, no GPU).

PARITY UNPINNED: the reference's code/tracking/sort.py needs filterpy, numba, skimage and a
scikit-learn module that no longer exists, so it cannot run here and it ships no fixtures.
These tests pin the published filter equations on hand-computed values and the bookkeeping
rules (min_hits, max_age, id numbering, output order) stated in sort.py:175-230."""
import numpy as np
import pytest

from facedetection_amd.parallel import feed_tracker
from facedetection_amd.tracking import sort


def box(cx, cy, w, h, s=0.9):
    return [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, s]


def test_state_conversions_round_trip():
    b = np.array([10.0, 20.0, 50.0, 100.0])
    z = sort.convert_bbox_to_z(b)
    np.testing.assert_allclose(z.ravel(), [30.0, 60.0, 40.0 * 80.0, 0.5])
    x = np.zeros((7, 1))
    x[:4] = z
    np.testing.assert_allclose(sort.convert_x_to_bbox(x), [b])
    assert sort.convert_x_to_bbox(x, score=0.7).shape == (1, 5)


def test_iou_has_no_plus_one_convention():
    assert sort.iou([0, 0, 10, 10], [5, 0, 15, 10]) == pytest.approx(50.0 / 150.0)
    assert sort.iou([0, 0, 10, 10], [20, 20, 30, 30]) == 0.0
    m = sort.iou_matrix(np.array([[0, 0, 10, 10, 1.0]]), np.array([[5, 0, 15, 10], [0, 0, 10, 10]]))
    np.testing.assert_allclose(m, [[1.0 / 3.0, 1.0]], rtol=1e-6)


def test_first_kalman_cycle_hand_computed():
    """Position/velocity pair of the centre: P0 = diag(10, 1e4), Q = diag(1, 0.01), R = 1.
    predict: P = [[10 + 1e4 + 1, 1e4], [1e4, 1e4 + 0.01]];  gain = P[:,0] / (P00 + 1)."""
    t = sort.KalmanBoxTracker(box(100, 100, 40, 40))
    t.predict()
    assert t.P[0, 0] == pytest.approx(10011.0) and t.P[0, 4] == pytest.approx(10000.0)
    assert t.P[4, 4] == pytest.approx(10000.01)
    t.update(box(106, 100, 40, 40))
    k_pos, k_vel = 10011.0 / 10012.0, 10000.0 / 10012.0
    assert t.x[0, 0] == pytest.approx(100.0 + k_pos * 6.0)
    assert t.x[4, 0] == pytest.approx(k_vel * 6.0)
    assert t.x[1, 0] == pytest.approx(100.0) and t.x[3, 0] == pytest.approx(1.0)
    np.testing.assert_allclose(t.P, t.P.T, atol=1e-9)
    # area / aspect use R = 10: area variance 10 + 1e4 + 1 -> gain 10011 / 10021
    t2 = sort.KalmanBoxTracker(box(0, 0, 10, 10))
    t2.predict()
    t2.update(box(0, 0, 20, 10))
    assert t2.x[2, 0] == pytest.approx(100.0 + 10011.0 / 10021.0 * 100.0)


def test_constant_velocity_track_converges():
    trk = sort.Sort()
    out = None
    for f in range(30):
        out = trk.update(np.array([box(100 + 4 * f, 200 + 2 * f, 50, 60)]))
    assert out.shape == (1, 5)
    np.testing.assert_allclose(out[0, :4], box(100 + 4 * 29, 200 + 2 * 29, 50, 60)[:4], atol=0.05)
    kt = trk.trackers[0]
    assert kt.x[4, 0] == pytest.approx(4.0, abs=0.02) and kt.x[5, 0] == pytest.approx(2.0, abs=0.02)


def test_bookkeeping_min_hits_max_age_ids_and_order():
    sort.KalmanBoxTracker.count = 0
    trk = sort.Sort(max_age=1, min_hits=3)
    a, b = box(100, 100, 40, 40), box(500, 300, 60, 60)
    out = trk.update(np.array([a, b]))
    # during the first min_hits frames every updated track is reported; newest first; ids from 1
    assert out[:, 4].tolist() == [2.0, 1.0]
    for _ in range(3):
        out = trk.update(np.array([a, b]))
    assert out[:, 4].tolist() == [2.0, 1.0]
    # a face appearing after the warm-up is reported once it has 3 consecutive hits after
    # the frame that created it
    c = box(900, 500, 50, 50)
    assert trk.update(np.array([a, b, c]))[:, 4].tolist() == [2.0, 1.0]
    assert trk.update(np.array([a, b, c]))[:, 4].tolist() == [2.0, 1.0]
    assert trk.update(np.array([a, b, c]))[:, 4].tolist() == [2.0, 1.0]
    assert trk.update(np.array([a, b, c]))[:, 4].tolist() == [3.0, 2.0, 1.0]
    # one missed frame keeps the track (max_age 1) but resets its streak; two drop it
    assert trk.update(np.array([a, b]))[:, 4].tolist() == [2.0, 1.0]
    assert len(trk.trackers) == 3
    trk.update(np.array([a, b]))
    assert len(trk.trackers) == 2
    # the same place later is a new identity
    for _ in range(4):
        out = trk.update(np.array([a, b, c]))
    assert sorted(out[:, 4].tolist()) == [1.0, 2.0, 4.0]


def test_empty_frames_and_low_overlap():
    trk = sort.Sort()
    assert trk.update(np.empty((0, 5))).shape == (0, 5)
    trk.update(np.array([box(100, 100, 40, 40)]))
    # a detection that overlaps the prediction by less than 0.3 starts a new track
    trk.update(np.array([box(135, 100, 40, 40)]))
    assert len(trk.trackers) == 2
    assert trk.update(np.zeros((0, 5), dtype=np.float32)).shape == (0, 5)


def test_assignment_is_optimal_not_greedy():
    dets = np.array([[0, 0, 10, 10, 1.0], [4, 0, 14, 10, 1.0]])
    trks = np.array([[5, 0, 15, 10], [-3, 0, 7, 10]], dtype=np.float64)
    m, ud, ut = sort.associate_detections_to_trackers(dets, trks)
    assert sorted(map(tuple, m.tolist())) == [(0, 1), (1, 0)] and ud.size == 0 and ut.size == 0


def test_feed_tracker_in_frame_order():
    sort.KalmanBoxTracker.count = 0
    frames = [np.array([box(100 + 3 * f, 100, 40, 40)], dtype=np.float32) for f in range(5)]
    frames.insert(2, np.zeros((0, 5), dtype=np.float32))
    outs = feed_tracker(sort.Sort(), frames)
    assert len(outs) == 6 and outs[2].shape == (0, 5)
    # the miss resets the streak after the warm-up: reported again at the third hit in a row
    assert [o.shape[0] for o in outs] == [1, 1, 0, 0, 0, 1]
    assert all(o[0, 4] == 1.0 for o in outs if o.shape[0])
