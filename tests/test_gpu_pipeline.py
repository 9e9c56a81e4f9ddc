"""Micro-batches over several HIP streams give exactly the single-stream results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_streams_match_single_stream_bit_exact():
    from facedetection_amd.detection import Detector
    from facedetection_amd.pipeline import FramePipeline
    from facedetection_amd.synthetic import synthetic_image
    frames = np.stack([synthetic_image(360, 640, seed=s) for s in range(7)])      # ragged last micro-batch
    det = Detector("synthetic", 0, precision="f16x3")
    try:
        want = det.detect_batch(frames)
    finally:
        det.close()
    for streams in (1, 2, 3):
        pipe = FramePipeline("synthetic", 0, streams=streams, precision="f16x3")
        try:
            got = pipe.detect(frames, micro_batch=2)
            again = pipe.detect(frames, micro_batch=3)
        finally:
            pipe.close()
        assert len(got) == len(want) == len(again) == 7
        for a, b, c in zip(got, want, again):
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(c, b)
