"""GPU parity of the network stages against the torch-CPU oracle (float64 yardstick)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import boxes as ob
from oracle import network as onet
from oracle import roipool as orp

pytestmark = pytest.mark.gpu

PROP = dict(pre_nms_top_n=1000, post_nms_top_n=10, nms_thresh=0.7, min_size=50)


@pytest.fixture(scope="module")
def ops():
    from facedetection_amd import ops
    return ops


@pytest.fixture(scope="module")
def params():
    from facedetection_amd.synthetic import synthetic_params
    return synthetic_params(0)


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def detector(params, request):
    from facedetection_amd.detection import Detector
    det = Detector(params, 0, precision=request.param)
    yield det
    det.close()


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ------------------------------------------------------------------ convolution operator
CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, extras
    (1, 37, 53, 64, 64, 1, 1, 0, "pre+bias+relu"),
    (2, 30, 41, 256, 128, 1, 1, 0, "pre+bias+relu"),
    (1, 45, 60, 64, 64, 3, 1, 1, "bias+relu"),
    (1, 45, 61, 128, 128, 3, 2, 1, "bias+relu"),
    (1, 33, 47, 128, 512, 1, 1, 0, "res"),
    (1, 33, 47, 256, 512, 1, 2, 0, "pre"),
    (1, 19, 23, 1024, 512, 3, 1, 1, "bias+relu"),
    (1, 19, 23, 512, 72, 1, 1, 0, "bias"),
    (3, 7, 7, 1024, 512, 1, 1, 0, "pre+bias+relu"),
    (3, 7, 7, 512, 512, 3, 2, 1, "bias+relu"),
    (1, 150, 200, 64, 256, 1, 1, 0, "res"),          # large M: 128x128 tiles
    (1, 140, 180, 64, 64, 3, 1, 1, "bias+relu"),     # large M, 64 outputs: 128x64 tiles
    (2, 250, 301, 64, 128, 3, 1, 1, "bias+relu"),    # grid fills the chip: 8-wave 256x128 tiles, ragged last row tile
    (1, 200, 301, 128, 264, 1, 1, 0, "res"),         # 8-wave 256x256 tiles, 8 channels in the last column tile
    (1, 180, 330, 32, 512, 3, 1, 1, "res"),          # 8-wave 256x256 tiles, K = 288
]


@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x3"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c[:8]) + "-" + c[8])
def test_conv2d_vs_float64(ops, case, precision):
    B, H, W, Cin, Cout, k, stride, pad, extras = case
    rng = np.random.default_rng(hash(case[:8]) % (2 ** 31))
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, Cin, k, k)) * np.sqrt(2.0 / (Cin * k * k))).astype(np.float32)
    bias = rng.normal(0, 0.5, Cout).astype(np.float32) if "bias" in extras else None
    ps = rng.uniform(0.5, 1.5, Cin).astype(np.float32) if "pre" in extras else None
    ph = rng.normal(0, 0.3, Cin).astype(np.float32) if "pre" in extras else None
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = rng.normal(0, 1, (B, Ho, Wo, Cout)).astype(np.float32) if "res" in extras else None
    y = ops.conv2d(x, w, stride=stride, pad=pad, bias=bias, pre_scale=ps, pre_shift=ph, relu="relu" in extras,
                   residual=res, precision=precision)
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    if ps is not None:
        xt = F.relu(xt * torch.from_numpy(ps).double()[None, :, None, None] + torch.from_numpy(ph).double()[None, :, None, None])
    ref = F.conv2d(xt, torch.from_numpy(w).double(), None if bias is None else torch.from_numpy(bias).double(),
                   stride=stride, padding=pad)
    if "relu" in extras:
        ref = F.relu(ref)
    ref = ref.permute(0, 2, 3, 1).numpy()
    if res is not None:
        ref = ref + res
    assert y.shape == ref.shape
    # float32 accumulation over K = Cin*k*k terms: error ~ 1e-7 * sqrt(K) * |terms|
    assert rel_err(y, ref) < 2e-6 * np.sqrt(Cin * k * k / 64.0) + 1e-6


# ------------------------------------------------------------------ whole graph, stage by stage
@pytest.fixture(scope="module")
def small_run(detector, params):
    from facedetection_amd.synthetic import synthetic_image
    img = synthetic_image(160, 224, seed=3)
    detector.keep_tensors(True)
    rois, cls_prob, bbox = detector.forward(img)
    taps = {n: detector.tensor(n) for n in ("conv0", "pool0", "stage1_unit1", "stage1_unit3", "stage2_unit4", "feat",
                                            "rpn_out", "rois", "roi_scores", "pooled", "stage4_unit3", "flat")}
    detector.keep_tensors(False)
    with torch.no_grad():
        o = {}
        data = onet.np2nd(img)
        dt = torch.float64
        x = onet.batch_norm(data.to(dt), params, "bn_data", dt, fix_gamma=True)
        x = onet.conv(x, params, "conv0", dt, stride=2, pad=3)
        x = F.relu(onet.batch_norm(x, params, "bn0", dt))
        o["conv0"] = x
        x = F.max_pool2d(x, 3, 2, 1)
        o["pool0"] = x
        x = onet.residual_unit(x, params, "stage1_unit1", dt, 1, False)
        o["stage1_unit1"] = x
        x = onet.residual_unit(x, params, "stage1_unit2", dt, 1, True)
        x = onet.residual_unit(x, params, "stage1_unit3", dt, 1, True)
        o["stage1_unit3"] = x
        x = onet.stage(x, params, 2, dt)
        o["stage2_unit4"] = x
        x = onet.stage(x, params, 3, dt)
        o["feat"] = x
        prob, bb, score = onet.rpn_head(x, params, dt)
        o["rpn_score"], o["rpn_bbox"], o["rpn_prob"] = score, bb, prob
    return dict(img=img, rois=rois, cls_prob=cls_prob, bbox=bbox, taps=taps, oracle=o)


@pytest.mark.parametrize("name,tol", [("conv0", 2e-6), ("pool0", 2e-6), ("stage1_unit1", 5e-6), ("stage1_unit3", 1e-5),
                                      ("stage2_unit4", 2e-5), ("feat", 5e-5)])
def test_backbone_activations(small_run, name, tol):
    got = small_run["taps"][name]
    ref = small_run["oracle"][name].permute(0, 2, 3, 1).numpy()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < tol


def test_rpn_head_outputs(small_run):
    got = small_run["taps"]["rpn_out"]               # [1,h,w,72] = bg(12) | fg(12) | deltas(48)
    score = small_run["oracle"]["rpn_score"].permute(0, 2, 3, 1).numpy()
    bbox = small_run["oracle"]["rpn_bbox"].permute(0, 2, 3, 1).numpy()
    assert np.abs(got[..., :24] - score).max() < 2e-4       # logits of magnitude ~10
    assert np.abs(got[..., 24:] - bbox).max() < 2e-5


def test_proposal_stage_inside_the_graph(small_run):
    """The oracle's proposal sequence on the GPU's own RPN outputs must pick the same boxes."""
    got = small_run["taps"]["rpn_out"]
    H, W = small_run["img"].shape[:2]
    score = torch.from_numpy(got[..., :24]).permute(0, 3, 1, 2)
    B, C2, h, w = score.shape
    prob = F.softmax(score.reshape(B, 2, 12 * h, w), dim=1).reshape(B, C2, h, w).numpy()
    bbox = np.ascontiguousarray(got[..., 24:].transpose(0, 3, 1, 2))
    rois, scores = ob.proposal(prob, bbox, np.array([H, W, 1.0], np.float32), **PROP)
    np.testing.assert_allclose(small_run["taps"]["roi_scores"].reshape(-1), scores, atol=2e-7)
    np.testing.assert_allclose(small_run["rois"], rois, rtol=1e-6, atol=1e-3)


def test_roi_pooling_inside_the_graph(small_run):
    feat = np.ascontiguousarray(small_run["taps"]["feat"].transpose(0, 3, 1, 2))
    ref = orp.roi_pooling(feat, small_run["rois"], (7, 7), 0.0625)
    np.testing.assert_array_equal(small_run["taps"]["pooled"], ref.transpose(0, 2, 3, 1))


def test_head_on_the_gpu_pooled_features(small_run, params):
    pooled = torch.from_numpy(np.ascontiguousarray(small_run["taps"]["pooled"].transpose(0, 3, 1, 2)))
    with torch.no_grad():
        cls_prob, bbox = onet.head(pooled, params, torch.float64)
    assert np.abs(small_run["cls_prob"] - cls_prob.numpy()).max() < 1e-4      # the 1e-4 score bar
    assert np.abs(small_run["bbox"] - bbox.numpy()).max() < 1e-4


# ------------------------------------------------------------------ end to end
@pytest.mark.parametrize("hw,seed", [((480, 640), 0), ((480, 640), 5), ((270, 480), 11), ((333, 517), 2)])
def test_detect_face_vs_oracle(detector, params, hw, seed):
    from facedetection_amd.synthetic import synthetic_image
    img = synthetic_image(hw[0], hw[1], seed=seed)
    dets = detector.detect_face(img)
    ref = onet.detect_face(img, params, PROP, dtype=torch.float32)
    assert dets.dtype == np.float32 and dets.ndim == 2 and dets.shape[1] == 5
    assert dets.shape == ref.shape, "different number of detections: %s vs %s" % (dets.shape, ref.shape)
    assert np.abs(dets[:, 4] - ref[:, 4]).max(initial=0) <= 1e-4          # scores within 1e-4
    assert np.abs(dets[:, :4] - ref[:, :4]).max(initial=0) <= 1.0          # rounded coords of boxes within 0.5 px
    raw = detector.detect_batch(img, do_round=False)[0]
    rois, cls_prob, deltas, _ = onet.forward(img, params, PROP, torch.float32)
    raw_ref = ob.postprocess(rois[:, 1:], cls_prob, deltas, img.shape[:2], do_round=False)
    assert np.abs(raw[:, :4] - raw_ref[:, :4]).max(initial=0) <= 0.5


def test_batch_equals_single_frames(detector):
    from facedetection_amd.synthetic import synthetic_image
    # 8 frames: enough ROI-head rows (8 x 10 rois x 16) for the larger tiles taken at large batches
    frames = np.stack([synthetic_image(256, 320, seed=s) for s in range(8)])
    batch = detector.detect_batch(frames)
    for b in range(8):
        single = detector.detect_face(frames[b])
        np.testing.assert_array_equal(batch[b], single)


def test_detect_returns_empty_landmarks(detector):
    from facedetection_amd.synthetic import synthetic_image
    boxes, lm = detector.detect(synthetic_image(128, 160, seed=1))
    assert lm.shape == (boxes.shape[0], 0)
