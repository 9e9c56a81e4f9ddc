"""Face crops (stick_boxes + resize_pad in one launch) against the NumPy oracle, through the
C ABI.  Integer/byte work: bit-exact."""
import numpy as np
import pytest

from oracle import crops as oc

pytestmark = pytest.mark.gpu


def frame(h=1080, w=1920, seed=3):
    from facedetection_amd.synthetic import synthetic_image
    return synthetic_image(h, w, seed=seed)


def some_boxes(rng, n, h, w):
    cx, cy = rng.uniform(0, w, n), rng.uniform(0, h, n)
    bw, bh = rng.uniform(8, 400, n), rng.uniform(8, 400, n)
    b = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2, rng.uniform(0, 1, n)], axis=1)
    return b


@pytest.mark.parametrize("des", [(112, 112), (96, 128), (160, 64)])
def test_tiles_and_rects_vs_oracle(des):
    from facedetection_amd import ops
    img = frame()
    rng = np.random.default_rng(5)
    boxes = some_boxes(rng, 64, *img.shape[:2])
    boxes[:8] = np.round(boxes[:8])            # integer-valued rows like the detector's output
    boxes[8] = [-30.7, -12.2, 55.5, 90.1, 1]   # over the top-left corner
    boxes[9] = [1800.0, 1000.0, 2100.0, 1300.0, 1]
    boxes[10] = [100, 100, 324, 324, 1]        # 224 + 2*22 = 268 px
    boxes[11] = [200, 200, 387, 387, 1]        # 187 + 2*19 = 225; -> 224 would be an exact 2x
    boxes[12] = [300, 300, 486, 486, 1]        # 186 + 2*19 = 224 -> exact 2x reduction to 112
    tiles, rects = ops.crop_resize_pad(img, boxes, des)
    ref_tiles, ref_rects = oc.crop_resize_pad(img, boxes, des)
    np.testing.assert_array_equal(rects, ref_rects)
    assert tiles.dtype == np.uint8 and tiles.shape == ref_tiles.shape
    np.testing.assert_array_equal(tiles, ref_tiles)


def test_float32_detector_rows_and_tracker_rows():
    from facedetection_amd import ops
    img = frame(480, 640, seed=4)
    dets = np.array([[12, 30, 80, 120, 0.99], [300, 200, 420, 330, 0.9]], dtype=np.float32)
    tiles, rects = ops.crop_resize_pad(img, dets)
    ref_tiles, ref_rects = oc.crop_resize_pad(img, dets, (112, 112))
    np.testing.assert_array_equal(rects, ref_rects)
    np.testing.assert_array_equal(tiles, ref_tiles)
    ids = np.array([[11.6, 29.4, 80.2, 120.9, 7.0]])          # fractional, from the Kalman state
    tiles, rects = ops.crop_resize_pad(img, ids)
    ref_tiles, ref_rects = oc.crop_resize_pad(img, ids, (112, 112))
    np.testing.assert_array_equal(rects, ref_rects)
    np.testing.assert_array_equal(tiles, ref_tiles)


def test_degenerate_and_empty():
    from facedetection_amd import ops
    img = frame(240, 320, seed=6)
    tiles, rects = ops.crop_resize_pad(img, np.zeros((0, 5)))
    assert tiles.shape == (0, 112, 112, 3) and rects.shape == (0, 4)
    boxes = np.array([[50.0, 60.0, 50.0, 90.0, 1.0],          # zero width
                      [400.0, 10.0, 500.0, 90.0, 1.0]])       # beyond the right edge
    tiles, rects = ops.crop_resize_pad(img, boxes)
    ref_tiles, ref_rects = oc.crop_resize_pad(img, boxes, (112, 112))
    np.testing.assert_array_equal(rects, ref_rects)
    np.testing.assert_array_equal(tiles, ref_tiles)
    assert not tiles.any()


def test_function_mirrors():
    from facedetection_amd import common
    img = frame(480, 640, seed=8)
    boxes = np.array([[12, 30, 80, 120, 1], [600, 400, 700, 500, 2]], dtype=np.float64)
    out, crops = common.stick_boxes(img, boxes)
    _, ref = oc.stick_boxes(img, boxes)
    assert out is img and len(crops) == 2
    for a, b in zip(crops, ref):
        np.testing.assert_array_equal(a, b)
    assert common.stick_boxes(img, np.zeros((0, 5)))[1] == []
    for c in ref:
        np.testing.assert_array_equal(common.resize_pad(c, (112, 112)), oc.resize_pad(c, (112, 112)))
    tall = img[:300, :100]
    np.testing.assert_array_equal(common.resize_pad(tall, (112, 112)), oc.resize_pad(tall, (112, 112)))


def test_detect_track_crop_pipeline():
    """The reference's loop body (code/data_processing/ProcessVideo.py:146-150) end to end."""
    from facedetection_amd import common
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    from facedetection_amd.tracking.sort import Sort
    det = Detector("synthetic", 0, precision="f16x3")
    tracker = Sort()
    try:
        n_ids = 0
        for f in range(3):
            img = synthetic_image(480, 640, seed=1)
            dets = det.detect_face(img)
            ids = tracker.update(dets)
            _, crops = common.stick_boxes(img, ids)
            assert len(crops) == ids.shape[0]
            n_ids += ids.shape[0]
            assert all(c.ndim == 3 and c.shape[2] == 3 for c in crops)
        assert n_ids == 3 * dets.shape[0]      # same frame thrice: every face tracked from frame 1
    finally:
        det.close()
