"""Host-side robustness: two host threads driving two detectors (own contexts, own streams) while
the event profiler is recording, and a context fed ever larger frames."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def params():
    from facedetection_amd.synthetic import synthetic_params
    return synthetic_params(0)


def test_two_threads_two_detectors_with_the_profiler_on(params):
    from facedetection_amd import profiling
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    frames = [np.stack([synthetic_image(256, 320, seed=10 * t + s) for s in range(4)]) for t in range(2)]
    dets = [Detector(params, 0, own_context=True, precision="f16x3") for _ in range(2)]
    try:
        expect = [dets[t].detect_batch(frames[t]) for t in range(2)]         # one after the other, no profiler
        results, errors = [None, None], []

        def work(t):
            try:
                with torch.cuda.device(0), torch.cuda.stream(torch.cuda.Stream()):
                    out = None
                    for _ in range(6):
                        out = dets[t].detect_batch(frames[t])
                    results[t] = out
            except Exception as e:                                             # surfaced below
                errors.append(e)

        profiling.begin()
        threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        rows = profiling.end()
    finally:
        for d in dets:
            d.close()
    assert not errors, errors
    for t in range(2):
        assert len(results[t]) == 4
        for a, b in zip(results[t], expect[t]):
            np.testing.assert_array_equal(a, b)
    # every launch of both threads was recorded: 12 passes of the same kernel sequence
    stem = [r for r in rows if r["kernel"] == "stem_pool"]
    assert stem and stem[0]["launches"] == 12 and all(r["ms"] > 0 for r in rows)


def test_growing_frames_do_not_accumulate_memory(params):
    """Every larger frame size replaces the activation buffers; the replaced ones are parked and
    must be released once they exceed the context's bound instead of living until destruction."""
    from facedetection_amd import runtime
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    det = Detector(params, 0, own_context=True, precision="f16x3")
    try:
        sizes = [(160 + 92 * i, 256 + 166 * i) for i in range(11)]            # 160x256 ... 1080x1916
        held = []
        for h, w in sizes:
            det.detect_face(synthetic_image(h, w, seed=1, faces=1))
            held.append(det._rt.scratch_bytes())
        final = held[-1]
        det2 = Detector(params, 0, own_context=True, precision="f16x3")
        try:
            det2.detect_face(synthetic_image(*sizes[-1], seed=1, faces=1))
            fresh = det2._rt.scratch_bytes()                                   # what the last size needs by itself
        finally:
            det2.close()
    finally:
        det.close()
    bound = 256 << 20
    assert final <= fresh * 1.6 + bound, (held, fresh)
    assert sum(held) > final                      # sanity: without releases `final` would be the sum of all steps
