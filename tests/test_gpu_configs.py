"""The remaining BASELINE.json configurations as parity cases: 4K frames and the dense-crowd
stress setting (every anchor into the RPN NMS, thousands of rois into the head)."""
import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle import cboxes
from oracle import network as onet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def params():
    from facedetection_amd.synthetic import synthetic_params
    return synthetic_params(0)


@pytest.fixture(scope="module")
def oracle_4k(params):
    """The float64 oracle's pass over the 4K test frame (computed once for both precisions)."""
    from facedetection_amd.synthetic import synthetic_image
    # four different quadrants: a 2 x 2 tiling of one image would give every detection three twins whose
    # scores differ by rounding noise only, and the order of the rows would be a coin toss
    q = [synthetic_image(1080, 1920, seed=21 + i) for i in range(4)]
    img = np.concatenate([np.concatenate(q[:2], axis=1), np.concatenate(q[2:], axis=1)], axis=0)     # 2160 x 3840
    cfg = dict(pre_nms_top_n=1000, post_nms_top_n=10, nms_thresh=0.7, min_size=50)
    ref, inter = onet.detect_face(img, params, cfg, dtype=torch.float64, return_all=True)
    return img, ref, inter


@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x3"])
def test_4k_frame_matches_oracle(params, oracle_4k, precision):
    from facedetection_amd.detection import Detector
    img, ref, inter = oracle_4k
    assert img.shape == (2160, 3840, 3)
    det = Detector(params, 0, precision=precision)
    try:
        dets = det.detect_face(img)
        feat = det.tensor("feat")
    finally:
        det.close()
    assert feat.shape == (1, 135, 240, 1024)
    # the stage-3 feature map of the ordinary (fused-kernel) pass against the float64 oracle
    ref_feat = inter["feat"].transpose(0, 2, 3, 1)
    assert np.abs(feat - ref_feat).max() / np.abs(ref_feat).max() < 5e-5
    assert dets.shape == ref.shape
    assert np.abs(dets[:, 4] - ref[:, 4]).max(initial=0) <= 1e-4
    assert np.abs(dets[:, :4] - ref[:, :4]).max(initial=0) <= 1.0


@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x3"])
def test_dense_crowd_settings(params, precision):
    """1080p, RPN_MIN_SIZE 12, no pre-NMS cut (97 920 candidates), 6 000 rois into stage 4."""
    from facedetection_amd.detection import Detector
    from facedetection_amd.symbol.config import config
    from facedetection_amd.synthetic import crowd_image
    from oracle import roipool as orp
    img = crowd_image(1080, 1920, seed=2, faces=5000)
    det = Detector(params, 0, precision=precision)
    old = (config.TEST.RPN_MIN_SIZE, config.TEST.RPN_PRE_NMS_TOP_N, config.TEST.RPN_POST_NMS_TOP_N)
    config.TEST.RPN_MIN_SIZE, config.TEST.RPN_PRE_NMS_TOP_N, config.TEST.RPN_POST_NMS_TOP_N = 12, 0, 6000
    try:
        det.reconfigure()
        rois, cls_prob, deltas = det.forward(img)
        rpn_out = det.tensor("rpn_out")
        feat = det.tensor("feat")
        dets = det.detect_face(img)
    finally:
        config.TEST.RPN_MIN_SIZE, config.TEST.RPN_PRE_NMS_TOP_N, config.TEST.RPN_POST_NMS_TOP_N = old
        det.close()
    assert rois.shape == (6000, 5) and cls_prob.shape == (6000, 2) and deltas.shape == (6000, 8)
    # RPN stage against the oracle's steps on the GPU's own head output (C restatement of the
    # greedy NMS: the NumPy one needs minutes at this size)
    A, (_, fh, fw, _) = 12, rpn_out.shape
    score = torch.from_numpy(rpn_out[..., :24]).permute(0, 3, 1, 2)
    prob = torch.softmax(score.reshape(1, 2, A * fh, fw), dim=1).reshape(1, 2 * A, fh, fw).numpy()
    fg = prob[0, A:].transpose(1, 2, 0).reshape(-1)
    d = rpn_out[0, :, :, 24:].reshape(-1, 4)
    anchors = ob.shifted_anchors(fh, fw, 16.0, ob.generate_anchors(16.0, (0.5, 1, 2), (4, 8, 16, 32)))
    boxes = ob.clip_boxes(ob.bbox_pred(anchors, d), (1080.0, 1920.0))
    ok = np.where((boxes[:, 2] - boxes[:, 0] + 1 >= 12) & (boxes[:, 3] - boxes[:, 1] + 1 >= 12))[0]
    order = ok[ob.score_order(fg[ok])]
    cand = np.concatenate([boxes[order], np.arange(order.size, 0, -1, dtype=np.float64)[:, None]], axis=1)
    keep = cboxes.nms(cand, 0.7)[:6000]
    assert len(keep) == 6000, "the stress input should leave at least 6000 proposals"
    np.testing.assert_allclose(rois[:, 1:], boxes[order][keep].astype(np.float32), rtol=1e-6, atol=1e-3)
    # Stage 4 + classifier at this batch size (faster_rcnn.py:103,137-160; 6 000 rois make the float32 pass take
    # its many-roi kernels: batched Winograd GEMMs on the 4x4 maps) against the float64 oracle: the first and
    # last rois and a stride through the middle, pooled by the oracle from the pass's own feature map
    pick = np.unique(np.concatenate([np.arange(64), np.arange(64, 5936, 46), np.arange(5936, 6000)]))
    assert pick.size >= 256
    pooled = orp.roi_pooling(feat.transpose(0, 3, 1, 2), rois[pick], (7, 7), 0.0625)
    with torch.no_grad():
        ref_prob, ref_deltas = onet.head(torch.from_numpy(pooled), params, torch.float64)
    assert np.abs(cls_prob[pick] - ref_prob.numpy()).max() <= 1e-4
    assert np.abs(deltas[pick] - ref_deltas.numpy()).max() <= 1e-4
    # final detections: sorted, mutually below the NMS threshold, scores from the head
    assert dets.shape[0] > 0 and (np.diff(dets[:, 4]) <= 0).all() and (dets[:, 4] >= 0.8).all()
    again = ob.nms(dets, 0.3) if dets.shape[0] < 3000 else cboxes.nms(dets, 0.3)
    # rounding moves coordinates by <= 0.5 px, so re-running NMS on the rounded boxes may drop a few
    assert len(again) >= 0.98 * dets.shape[0]
    ref = ob.postprocess(rois[:, 1:], cls_prob, deltas, (1080, 1920))
    assert ref.shape == dets.shape
    assert np.abs(ref[:, 4] - dets[:, 4]).max() <= 1e-6 and np.abs(ref[:, :4] - dets[:, :4]).max() <= 1.0


def test_micro_batch_beyond_31_bit_offsets_matches_default_path(params):
    """18 frames of 1080p make the stem output and the stage-1 tensors larger than 2^31 bytes (2.39 GB; 16 frames
    are 2.12 GB and still fit), which moves those layers of the f16x3 pass from the buffer-addressed kernels to the
    pointer-addressed ones.  The graph outputs must agree with the same frames run as two calls of 9 (different
    kernels, same arithmetic up to summation order)."""
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    base = [synthetic_image(1080, 1920, seed=100 + s) for s in range(6)]
    frames = np.stack([np.roll(base[s % 6], 53 * (s // 6), axis=1) for s in range(18)])
    det = Detector(params, 0, precision="f16x3")
    try:
        big = det.forward(frames)
        halves = [det.forward(frames[i:i + 9]) for i in (0, 9)]
        dets18 = det.detect_batch(frames)
        dets9 = det.detect_batch(frames[:9]) + det.detect_batch(frames[9:])
    finally:
        det.close()
    R = 10
    for k, tol in ((0, 0.05), (1, 1e-5), (2, 1e-4)):
        ref = np.concatenate([h[k] for h in halves])
        if k == 0:                                    # rois carry the batch index of their own call
            ref[9 * R:, 0] += 9
        assert big[k].shape == ref.shape
        assert np.abs(big[k] - ref).max() <= tol, (k, np.abs(big[k] - ref).max())
    assert len(dets18) == 18
    for a, b in zip(dets18, dets9):
        assert a.shape == b.shape
        assert np.abs(a[:, 4] - b[:, 4]).max(initial=0) <= 1e-5 and np.abs(a[:, :4] - b[:, :4]).max(initial=0) <= 1.0


def test_float32_call_beyond_one_pass_equals_single_frames(params, monkeypatch):
    """A float32 call whose tensors would not fit the buffer-addressed kernels' 2^31-byte reach is run in
    several passes by the library (here forced: at most 2 frames per pass, 5 frames -> 2 + 2 + 1); a frame's
    detections must equal, bit for bit, what it gives alone, and the status block covers all passes."""
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    frames = np.stack([synthetic_image(360, 640, seed=40 + s) for s in range(5)])
    det = Detector(params, 0, precision="f32")
    try:
        alone = [det.detect_face(f, thresh=0.5) for f in frames]
        monkeypatch.setenv("FD_F32_PASS_FRAMES", "2")
        f = torch.from_numpy(frames).cuda()
        dets, n, status = det.detect_batch_device(f, thresh=0.5)
        torch.cuda.synchronize()
        n = n.cpu().numpy()
        dets = dets.cpu().numpy()
        scaled = det.detect_scaled_batch(frames, target_size=300, max_size=500, thresh=0.5)
        monkeypatch.delenv("FD_F32_PASS_FRAMES")
        scaled_one_pass = det.detect_scaled_batch(frames, target_size=300, max_size=500, thresh=0.5)
    finally:
        det.close()
    assert sum(len(a) for a in alone) > 0 and status.cpu().tolist() == [0, 0]
    for i in range(5):
        np.testing.assert_array_equal(dets[i, :n[i]], alone[i])
    for a, b in zip(scaled[0], scaled_one_pass[0]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(scaled[1], scaled_one_pass[1]):
        np.testing.assert_array_equal(a, b)


def test_float32_call_of_18_frames_of_1080p(params):
    """18 frames of 1080p in float32: the stage-1 tensors of more than 16 frames exceed 2^31 bytes, so the call
    takes two passes (16 + 2); results equal those of the frames run as two calls."""
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image
    base = [synthetic_image(1080, 1920, seed=300 + s) for s in range(3)]
    frames = np.stack([np.roll(base[s % 3], 37 * s, axis=1) for s in range(18)])
    det = Detector(params, 0, precision="f32")
    try:
        whole = det.detect_batch(frames, thresh=0.5)
        parts = det.detect_batch(frames[:16], thresh=0.5) + det.detect_batch(frames[16:], thresh=0.5)
    finally:
        det.close()
    assert len(whole) == 18 and sum(len(a) for a in whole) > 0
    for a, b in zip(whole, parts):
        np.testing.assert_array_equal(a, b)
