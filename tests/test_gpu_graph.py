"""A detection call recorded as a HIP graph (fd_detect_graph_create) must give, on every replay, exactly
what the ordinary call gives for the frames in its input buffer -- the same kernels run with the same
arguments -- and must refuse to replay once the library moved a buffer it recorded."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_replays_equal_ordinary_calls(precision):
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    frames = [synthetic_image(360, 640, seed=60 + s) for s in range(4)]
    det = Detector("synthetic:0", 0, precision=precision)
    try:
        want = [det.detect_face(f, thresh=0.5) for f in frames]
        rec = det.record(1, 360, 640, thresh=0.5)
        got = [rec.detect(f[None])[0] for f in frames]
        again = rec.detect(frames[0][None])[0]                 # the buffers keep nothing from the previous replay
        status = rec.status.cpu().tolist()
        rec.close()
        # a batch, recorded after the single-frame pass (larger buffers: the first recording would be stale now)
        batch = np.stack(frames[:3])
        rec3 = det.record(3, 360, 640, thresh=0.5)
        got3 = rec3.detect(batch)
        rec3.close()
    finally:
        det.close()
    assert sum(len(w) for w in want) > 0 and status == [0, 0]
    for w, g in zip(want, got):
        np.testing.assert_array_equal(g, w)
    np.testing.assert_array_equal(again, want[0])
    for w, g in zip(want[:3], got3):
        np.testing.assert_array_equal(g, w)


def test_single_1080p_frame_replayed():
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    frames = [synthetic_image(1080, 1920, seed=70 + s) for s in range(2)]
    det = Detector("synthetic:0", 0, precision="f16x3")
    try:
        want = [det.detect_face(f) for f in frames]
        got = [det.detect_face_replayed(f) for f in frames]
        got += [det.detect_face_replayed(frames[0])]
    finally:
        det.close()
    for w, g in zip(want + [want[0]], got):
        np.testing.assert_array_equal(g, w)


def test_stale_recording_is_refused_and_recorded_again():
    from facedetection_amd import _lib
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    small = synthetic_image(256, 320, seed=80)
    det = Detector("synthetic:0", 0, own_context=True, precision="f16x3")       # a context whose buffers start empty
    try:
        want = det.detect_face(small, thresh=0.5)
        rec = det.record(1, 256, 320, thresh=0.5)
        np.testing.assert_array_equal(rec.detect(small[None])[0], want)
        det.detect_face_replayed(small, thresh=0.5)
        det.detect_face(synthetic_image(720, 1280, seed=81))   # larger shape: the library re-allocates
        with pytest.raises(_lib.FaceDetError, match="re-allocated"):
            rec.launch()
        assert rec.stale
        rec.close()
        np.testing.assert_array_equal(det.detect_face_replayed(small, thresh=0.5), want)   # records again
    finally:
        det.close()
    torch.cuda.synchronize()
