"""An MXNet checkpoint on disk as the detector's model (code/detection/detection.py:28, :81-88): the synthetic
parameter table written as ``prefix-0000.params`` by checkpoint.save_params (NDArray-list container, V2 records,
"arg:" / "aux:" key prefixes), read back by ``Detector(prefix, ctx, epoch)`` and ``init_detector(prefix, epoch, gpu)``;
detections must equal those of a detector built from the same table in memory, bit for bit.
PARITY UNPINNED for the container format: no file written by MXNet itself is available offline."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AUX_SUFFIXES = ("_moving_mean", "_moving_var")


@pytest.fixture(scope="module")
def prefix(tmp_path_factory):
    from facedetection_amd import checkpoint as ck
    from facedetection_amd.synthetic import synthetic_params
    params = synthetic_params(0)
    arg = {k: v for k, v in params.items() if not k.endswith(AUX_SUFFIXES)}
    aux = {k: v for k, v in params.items() if k.endswith(AUX_SUFFIXES)}
    assert arg and aux
    pre = str(tmp_path_factory.mktemp("ckpt") / "mxnet-face-fr50")
    ck.save_params(pre + "-0000.params", arg, aux, version=2)
    return pre, params


def test_detector_from_a_params_file_equals_the_in_memory_model(prefix):
    from facedetection_amd.detection import Detector, detect_face, init_detector
    from facedetection_amd.symbol.config import config
    from facedetection_amd.synthetic import synthetic_image
    pre, params = prefix
    frames = [synthetic_image(360, 640, seed=70 + i) for i in range(3)]
    saved_min_size = config.TEST.RPN_MIN_SIZE
    mem = Detector(params, 0)
    disk = Detector(pre, 0, 0)
    try:
        assert config.TEST.RPN_MIN_SIZE == 50                      # detection.py:27
        want = [mem.detect_face(f, thresh=0.3) for f in frames]
        got = [disk.detect_face(f, thresh=0.3) for f in frames]
    finally:
        disk.close()
    assert sum(len(w) for w in want) > 0
    for w, g in zip(want, got):
        np.testing.assert_array_equal(g, w)
    # the free-function form (detection.py:81-88 and its callers): min-size 24
    try:
        arg_params, aux_params, sym, ctx = init_detector(pre, 0, 0)
        try:
            assert config.TEST.RPN_MIN_SIZE == 24
            config_saved = config.TEST.RPN_MIN_SIZE
            mem.reconfigure()                                      # the in-memory detector at min-size 24 as well
            for f in frames:
                np.testing.assert_array_equal(detect_face(ctx, sym, arg_params, aux_params, f, thresh=0.3),
                                              mem.detect_face(f, thresh=0.3))
            assert config.TEST.RPN_MIN_SIZE == config_saved
        finally:
            sym.close()
    finally:
        config.TEST.RPN_MIN_SIZE = saved_min_size
        mem.close()


def test_missing_or_truncated_checkpoint_is_refused(prefix, tmp_path):
    from facedetection_amd.detection import Detector
    pre, _ = prefix
    with pytest.raises((OSError, ValueError)):
        Detector(str(tmp_path / "nothing-here"), 0, 0)
    blob = open(pre + "-0000.params", "rb").read()
    cut = tmp_path / "cut-0000.params"
    cut.write_bytes(blob[:len(blob) // 2])
    with pytest.raises(ValueError):
        Detector(str(tmp_path / "cut"), 0, 0)
