"""Parity of the kernels an ordinary pass runs (fused stem+pool, whole-unit kernels, fused
bottleneck tails, pooling straight into the split form, the 8-wave tiles) at the sizes of the
benchmark configurations: a 1080p frame against the float64 oracle tensor by tensor, the default
detector on one 1080p frame, and 32 frames of 1080p through the two-stream pipeline."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import network as onet

pytestmark = pytest.mark.gpu

PROP = dict(pre_nms_top_n=1000, post_nms_top_n=10, nms_thresh=0.7, min_size=50)
H, W = 1080, 1920
TAPS = ("stage1_unit1", "stage1_unit3", "stage2_unit4", "stage3_unit1", "feat", "rpn_out", "rois", "cls_prob", "bbox_pred")


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def params():
    from facedetection_amd.synthetic import synthetic_params
    return synthetic_params(0)


@pytest.fixture(scope="module")
def frame():
    from facedetection_amd.synthetic import synthetic_image
    return synthetic_image(H, W, seed=1)


@pytest.fixture(scope="module")
def oracle64(params, frame):
    """Float64 evaluation of the graph on the 1080p frame, unit outputs kept."""
    dt = torch.float64
    o = {}
    with torch.no_grad():
        x = onet.batch_norm(onet.np2nd(frame).to(dt), params, "bn_data", dt, fix_gamma=True)
        x = F.relu(onet.batch_norm(onet.conv(x, params, "conv0", dt, stride=2, pad=3), params, "bn0", dt))
        x = F.max_pool2d(x, 3, 2, 1)
        x = onet.residual_unit(x, params, "stage1_unit1", dt, 1, False)
        o["stage1_unit1"] = x.permute(0, 2, 3, 1).numpy()
        x = onet.residual_unit(x, params, "stage1_unit2", dt, 1, True)
        x = onet.residual_unit(x, params, "stage1_unit3", dt, 1, True)
        o["stage1_unit3"] = x.permute(0, 2, 3, 1).numpy()
        x = onet.stage(x, params, 2, dt)
        o["stage2_unit4"] = x.permute(0, 2, 3, 1).numpy()
        x = onet.residual_unit(x, params, "stage3_unit1", dt, 2, False)
        o["stage3_unit1"] = x.permute(0, 2, 3, 1).numpy()
        for j in range(2, 7):
            x = onet.residual_unit(x, params, "stage3_unit%d" % j, dt, 1, True)
        o["feat"] = x.permute(0, 2, 3, 1).numpy()
        _, bbox, score = onet.rpn_head(x, params, dt)
        o["rpn_score"] = score.permute(0, 2, 3, 1).numpy()
        o["rpn_bbox"] = bbox.permute(0, 2, 3, 1).numpy()
    rois, cls_prob, deltas, _ = onet.forward(frame, params, PROP, dt)
    o.update(rois=rois, cls_prob=cls_prob, bbox_pred=deltas)
    return o


@pytest.fixture(scope="module", params=["f16x3", "f32", "bf16x3"])
def production_run(params, frame, request):
    """One ordinary pass (kernel selection untouched) with the tensors it materialises kept."""
    from facedetection_amd.detection import Detector
    det = Detector(params, 0, precision=request.param, on_clamp="raise")
    try:
        det.keep_tensors(True, production=True)
        rois, cls_prob, bbox = det.forward(frame)
        taps = {n: det.tensor(n) for n in TAPS}
        det.keep_tensors(False)
        # the same pass without any keeping must give the same outputs bit for bit
        rois2, cls2, bbox2 = det.forward(frame)
        feat2 = det.tensor("feat")
    finally:
        det.close()
    np.testing.assert_array_equal(rois, rois2)
    np.testing.assert_array_equal(cls_prob, cls2)
    np.testing.assert_array_equal(bbox, bbox2)
    np.testing.assert_array_equal(taps["feat"], feat2)
    return dict(precision=request.param, taps=taps, rois=rois, cls_prob=cls_prob, bbox=bbox)


@pytest.mark.parametrize("name,tol", [("stage1_unit1", 5e-6), ("stage1_unit3", 1e-5), ("stage2_unit4", 2e-5),
                                      ("stage3_unit1", 3e-5), ("feat", 5e-5)])
def test_backbone_of_an_ordinary_pass_at_1080p(production_run, oracle64, name, tol):
    got, ref = production_run["taps"][name], oracle64[name]
    assert got.shape == ref.shape
    assert rel_err(got, ref) < tol


def test_rpn_head_of_an_ordinary_pass_at_1080p(production_run, oracle64):
    got = production_run["taps"]["rpn_out"]               # [1,68,120,72] = bg(12) | fg(12) | deltas(48)
    assert got.shape == (1, 68, 120, 72)
    assert np.abs(got[..., :24] - oracle64["rpn_score"]).max() < 2e-4       # logits of magnitude ~10
    assert np.abs(got[..., 24:] - oracle64["rpn_bbox"]).max() < 2e-5


def test_graph_outputs_of_an_ordinary_pass_at_1080p(production_run, oracle64):
    assert np.abs(production_run["rois"] - oracle64["rois"]).max() <= 0.05
    assert np.abs(production_run["cls_prob"] - oracle64["cls_prob"]).max() <= 1e-4      # the 1e-4 score bar
    assert np.abs(production_run["bbox"] - oracle64["bbox_pred"]).max() <= 1e-4
    np.testing.assert_array_equal(production_run["taps"]["cls_prob"].reshape(-1, 2), production_run["cls_prob"])


@pytest.mark.parametrize("precision", ["f16x3", "f32", "bf16x3"])
def test_default_detector_on_one_1080p_frame(params, frame, precision):
    """BASELINE configuration 2: Detector defaults (thresh 0.8, nms 0.3, min size 50, 1000 / 10 proposals)."""
    from facedetection_amd.detection import Detector
    from oracle import boxes as ob
    det = Detector(params, 0, precision=precision)
    try:
        dets = det.detect_face(frame)
        raw = det.detect_batch(frame, do_round=False)[0]
    finally:
        det.close()
    rois, cls_prob, deltas, _ = onet.forward(frame, params, PROP, torch.float64)
    ref = ob.postprocess(rois[:, 1:], cls_prob, deltas, (H, W))
    raw_ref = ob.postprocess(rois[:, 1:], cls_prob, deltas, (H, W), do_round=False)
    assert dets.shape == ref.shape and dets.shape[0] > 0
    assert np.abs(dets[:, 4] - ref[:, 4]).max() <= 1e-4
    assert np.abs(raw[:, :4] - raw_ref[:, :4]).max() <= 0.5
    assert np.abs(dets[:, :4] - ref[:, :4]).max() <= 1.0           # rounded coordinates of boxes within 0.5 px


def _frames_1080p(n, seed=7):
    """n distinct 1080p frames from 4 generated ones (rolled and re-noised), uint8 on the GPU."""
    from facedetection_amd.synthetic import synthetic_image
    base = torch.from_numpy(np.stack([synthetic_image(H, W, seed=seed * 100 + i) for i in range(4)])).cuda()
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    for i in range(n):
        f = torch.roll(base[i % 4], shifts=(7 * i % H, 13 * i % W), dims=(0, 1)).to(torch.int16)
        f = f + torch.randint(-6, 7, f.shape, generator=g, device="cuda", dtype=torch.int16)
        out[i] = f.clamp_(0, 255).to(torch.uint8)
    return out


@pytest.mark.parametrize("precision", ["f16x3", "f32", "bf16x3"])
def test_pipeline_of_32_frames_equals_frame_by_frame_detection(params, precision):
    """BASELINE configuration 3 at a size the test can afford: 32 frames of 1080p through
    FramePipeline(streams=2, micro_batch=8) -- the benchmark's code path -- against
    Detector.detect_face on every frame alone, bit for bit, plus an oracle check of two frames."""
    from facedetection_amd.detection import Detector
    from facedetection_amd.pipeline import FramePipeline
    from oracle import boxes as ob
    frames = _frames_1080p(32)
    pipe = FramePipeline(params, 0, streams=2, precision=precision)
    try:
        got = pipe.detect(frames, micro_batch=8)
    finally:
        pipe.close()
    det = Detector(params, 0, precision=precision)
    try:
        single = [det.detect_face(frames[i]) for i in range(32)]
    finally:
        det.close()
    assert len(got) == 32 and sum(len(g) for g in got) > 0
    for i in range(32):
        np.testing.assert_array_equal(got[i], single[i], err_msg="frame %d" % i)
    for i in (0, 19):
        img = frames[i].cpu().numpy()
        rois, cls_prob, deltas, _ = onet.forward(img, params, PROP, torch.float64)
        ref = ob.postprocess(rois[:, 1:], cls_prob, deltas, (H, W))
        assert got[i].shape == ref.shape
        assert np.abs(got[i][:, 4] - ref[:, 4]).max(initial=0) <= 1e-4
        assert np.abs(got[i][:, :4] - ref[:, :4]).max(initial=0) <= 1.0


@pytest.mark.parametrize("precision,streams", [("f32", 1), ("f16x3", 2)])
def test_benchmark_batch_of_256_frames_equals_frame_by_frame_detection(params, precision, streams):
    """BASELINE configuration 3 at its own size and in the benchmark's own arrangement: 256 frames of 1080p
    in micro-batches of 16 (float32 on one stream, f16x3 on two) against Detector.detect_face on every one
    of the 256 frames alone, bit for bit -- a frame's rows depend neither on its micro-batch nor on the stream."""
    from facedetection_amd.detection import Detector
    from facedetection_amd.pipeline import FramePipeline
    frames = _frames_1080p(256)
    pipe = FramePipeline(params, 0, streams=streams, precision=precision)
    try:
        got = pipe.detect(frames, micro_batch=16)
    finally:
        pipe.close()
    det = Detector(params, 0, precision=precision)
    try:
        single = [det.detect_face(frames[i]) for i in range(256)]
    finally:
        det.close()
    assert len(got) == 256 and sum(len(g) for g in got) > 256
    for i in range(256):
        np.testing.assert_array_equal(got[i], single[i], err_msg="frame %d" % i)
