"""The command-line loop (resize + detect + map back) and the video helpers on the GPU."""
import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle import network as onet
from oracle import resize as oresize

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def params():
    from facedetection_amd.synthetic import synthetic_params
    return synthetic_params(0)


def test_resize_with_channel_swap_bit_exact():
    import ctypes as C
    from facedetection_amd import _lib, runtime
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (2, 90, 160, 3), dtype=np.uint8)
    for scale in (0.5, 0.73, 1.4):
        dh, dw = oresize.dst_size(90, 160, scale, scale)
        src = runtime.to_device(img, torch.uint8)
        dst = runtime.empty((2, dh, dw, 3), torch.uint8)
        _lib.check(_lib.lib().fd_resize_cvt_u8(runtime.context().handle, runtime.ptr(src), 2, 90, 160, 3, scale, scale,
                                               1, runtime.ptr(dst), dh, dw, runtime.stream_ptr()))
        got = dst.cpu().numpy()
        for b in range(2):
            np.testing.assert_array_equal(got[b], oresize.resize_linear_u8(img[b, :, :, ::-1], scale, scale))


def test_scaled_batch_matches_oracle_loop(params):
    """detection.py:105-148 per frame on the CPU restatement against one batched call."""
    from facedetection_amd.detection import Detector
    from facedetection_amd.symbol.config import config
    from facedetection_amd.synthetic import synthetic_image
    frames = np.stack([synthetic_image(360, 640, seed=s) for s in (9, 10)])
    det = Detector(params, 0, precision="f16x3")
    config.TEST.RPN_MIN_SIZE = 24
    det.reconfigure()
    try:
        dets, boxes, scale = det.detect_scaled_batch(frames, 300, 500)
    finally:
        config.TEST.RPN_MIN_SIZE = 50
        det.close()
    assert scale == 500.0 / 640.0             # the long side caps the factor
    cfg = dict(pre_nms_top_n=1000, post_nms_top_n=10, nms_thresh=0.7, min_size=24)
    for b in range(2):
        rgb = frames[b][:, :, ::-1]
        img, s = oresize.resize(rgb, 300, 500)
        rois, cls, deltas, _ = onet.forward(img, params, cfg, torch.float32, im_scale=s)
        ref = ob.postprocess(rois[:, 1:], cls, deltas, img.shape[:2], do_round=False)
        assert dets[b].shape == ref.shape
        assert np.abs(dets[b][:, 4] - ref[:, 4]).max(initial=0) <= 1e-4
        assert np.abs(dets[b][:, :4] - ref[:, :4]).max(initial=0) <= 0.5
        # the mapping back is exact integer work on the GPU's own rows
        want = np.array([[int(round(float(v) / s)) for v in row[:4]] for row in dets[b]], dtype=np.int32).reshape(-1, 4)
        np.testing.assert_array_equal(boxes[b], want)


def test_cli_main_writes_face_regions(tmp_path, capsys):
    from facedetection_amd import video
    from facedetection_amd.synthetic import synthetic_image
    frames = np.stack([synthetic_image(480, 640, seed=s) for s in (1, 2, 3)])
    np.save(str(tmp_path / "clip.npy"), frames)
    out = tmp_path / "faces"
    rc = video.main(["--frames", str(tmp_path / "clip.npy"), "--prefix", "synthetic", "--out", str(out), "--batch", "2",
                     "--scale", "480", "--max-scale", "800"])
    assert rc == 0
    text = capsys.readouterr().out
    assert "3 frames" in text and text.count("time cost is:") == 2
    n = int(text.strip().splitlines()[-1].split(",")[1].split()[0])
    assert len(list(out.glob("*.ppm"))) <= n
    for f in out.glob("*.ppm"):
        assert video.read_ppm(str(f)).size > 0


def test_compress_video_keeps_face_stretches(params):
    from facedetection_amd import video
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    from oracle import video as ov
    a = synthetic_image(240, 320, seed=1)
    b = np.zeros_like(a)
    det = Detector(params, 0, precision="f16x3")
    try:
        # random weights see "faces" everywhere: put the score threshold between the two frames
        sa = det.detect_face(a, thresh=0.0)[:, 4].max(initial=0.0)
        sb = det.detect_face(b, thresh=0.0)[:, 4].max(initial=0.0)
        assert sa != sb
        thresh = float(sa + sb) / 2.0
        hit, miss = (a, b) if sa > sb else (b, a)
        seq = [miss] * 6 + [hit] * 2 + [miss] * 12 + [hit] + [miss] * 3
        kept = list(video.compress_video(seq, det, batch=4, buffer_size=4, thresh=thresh))
    finally:
        det.close()
    want = ov.written_frames([f is hit for f in seq], 4)
    assert 0 < len(want) < len(seq) and len(kept) == len(want)
    for k, i in zip(kept, want):
        np.testing.assert_array_equal(k, seq[i])
