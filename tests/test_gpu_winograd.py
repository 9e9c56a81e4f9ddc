"""precision="f32" runs the 3x3 convolutions of the 256-channel stage and of the RPN by Winograd's minimal
filtering: F(4x4, 3x3) by default (a quarter of the multiplies of the direct form), F(2x2, 3x3) with
FD_WINOGRAD=2 (4/9 of them, transforms made of additions only).  Checked against the direct float32
convolutions of a layer-by-layer pass (same weights) on feature maps with odd and even sides, and against
float64 arithmetic on the RPN layer alone and on the operator."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [(1, 37, 41), (3, 64, 96), (2, 333, 517), (1, 270, 480), (2, 1080, 1920)]


@pytest.mark.parametrize("tile", ["4", "2"])
def test_winograd_rpn_convolution_matches_the_direct_float32_kernel(monkeypatch, tile):
    from facedetection_amd.detection import Detector
    monkeypatch.setenv("FD_WINOGRAD", tile)
    lay = Detector("synthetic:0", 0, own_context=True, precision="f32")
    new = Detector("synthetic:0", 0, own_context=True, precision="f32")
    lay.keep_tensors(True)                       # layer by layer: direct convolution
    new.keep_tensors(True, production=True)      # ordinary pass: Winograd
    rng = np.random.default_rng(5)
    try:
        for (b, h, w) in SIZES:
            frames = rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)
            lay.forward(frames)
            new.forward(frames)
            feat_a, feat_b = lay.tensor("feat"), new.tensor("feat")
            scale = float(np.abs(feat_a).max())
            assert float(np.abs(feat_a - feat_b).max()) <= 1.5e-5 * scale, (b, h, w)
            want, got = lay.tensor("rpn_out"), new.tensor("rpn_out")
            assert want.shape == got.shape
            scale = float(np.abs(want).max())
            # the inputs already differ by the rounding noise checked above; the two forms of the
            # convolution add their own (a 9216-term sum each)
            assert float(np.abs(got - want).max()) <= 4e-5 * scale, (b, h, w)
    finally:
        lay.close()
        new.close()


def test_winograd_rpn_convolution_against_float64(monkeypatch):
    """The layer alone against float64 arithmetic on the same float32 input: the error of the Winograd
    form stays within a small multiple of the direct float32 kernel's own error."""
    from facedetection_amd.detection import Detector
    from facedetection_amd import synthetic
    params = synthetic.synthetic_params(0) if hasattr(synthetic, "synthetic_params") else None
    if params is None:
        pytest.skip("synthetic parameter table not exposed")
    w = torch.from_numpy(np.asarray(params["rpn_conv_3x3_weight"], dtype=np.float32)).double()
    bias = torch.from_numpy(np.asarray(params["rpn_conv_3x3_bias"], dtype=np.float32)).double()
    wo = torch.from_numpy(np.concatenate([np.asarray(params["rpn_cls_score_weight"], dtype=np.float32).reshape(-1, 512),
                                          np.asarray(params["rpn_bbox_pred_weight"], dtype=np.float32).reshape(-1, 512)])).double()
    bo = torch.from_numpy(np.concatenate([np.asarray(params["rpn_cls_score_bias"], dtype=np.float32),
                                          np.asarray(params["rpn_bbox_pred_bias"], dtype=np.float32)])).double()
    errs = {}
    rng = np.random.default_rng(6)
    frames = rng.integers(0, 256, size=(1, 333, 517, 3), dtype=np.uint8)
    for mode in ("4", "2", "0"):
        monkeypatch.setenv("FD_WINOGRAD", mode)
        det = Detector("synthetic:0", 0, own_context=True, precision="f32")
        det.keep_tensors(True, production=True)
        try:
            det.forward(frames)
            feat = torch.from_numpy(det.tensor("feat")).double().permute(0, 3, 1, 2)
            hid = torch.relu(torch.nn.functional.conv2d(feat, w, bias, padding=1))
            ref = torch.nn.functional.conv2d(hid, wo[:, :, None, None], bo).permute(0, 2, 3, 1).numpy()
            got = det.tensor("rpn_out")
            errs[mode] = float(np.abs(got - ref).max()) / float(np.abs(ref).max())
        finally:
            det.close()
    assert errs["0"] <= 5e-6, errs
    assert errs["2"] <= 4 * max(errs["0"], 5e-7), errs
    assert errs["4"] <= 6 * max(errs["0"], 5e-7), errs


@pytest.mark.parametrize("b,h,w,cin,cout", [(1, 1, 1, 32, 8), (2, 5, 7, 32, 72), (3, 8, 8, 64, 64), (1, 37, 51, 96, 40),
                                            (2, 68, 120, 256, 256),
                                            # the ROI head's shape with many rois (4x4 maps, one mostly zero-filled
                                            # 6x6 patch each, 512 channels): the float32 pass runs it once a call
                                            # carries 1 024 rois or more
                                            (1024, 4, 4, 512, 512)])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("form", ["winograd", "winograd4"])
def test_winograd_operator_against_float64(b, h, w, cin, cout, relu, form):
    """The operator alone (fd_conv2d_winograd through the C ABI) on odd and even map sizes, a single pixel,
    channel counts that leave GEMM tiles ragged; float64 torch convolution as the reference, and the
    direct float32 kernel next to it: the Winograd forms may carry a few times its rounding noise."""
    from facedetection_amd import ops
    rng = np.random.default_rng(b * 1000 + h)
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double(),
                                     torch.from_numpy(bias).double(), padding=1)
    if relu:
        ref = torch.relu(ref)
    ref = ref.permute(0, 2, 3, 1).numpy()
    got = ops.conv2d(x, wt, stride=1, pad=1, bias=bias, relu=relu, precision=form)
    direct = ops.conv2d(x, wt, stride=1, pad=1, bias=bias, relu=relu, precision="f32")
    assert got.shape == ref.shape
    scale = max(float(np.abs(ref).max()), 1.0)
    e_w, e_d = float(np.abs(got - ref).max()) / scale, float(np.abs(direct - ref).max()) / scale
    assert e_d <= 3e-6, e_d
    # F(2x2): transforms of additions only; F(4x4): fractional entries up to 3.4, about three times the noise
    assert e_w <= (max(4 * e_d, 2e-6) if form == "winograd" else max(8 * e_d, 5e-6)), (e_w, e_d)


def test_split_precision_winograd_against_float64_and_the_direct_form(monkeypatch):
    """precision="f16x3" runs the RPN's 3x3 convolution and those of the 256-channel stage as F(4x4, 3x3) on
    fp16 hi+lo pairs (FD_WINOGRAD_SPLIT=0: direct).  The RPN layer against float64 arithmetic on the pass's own
    input, next to the direct split-precision kernel, and the stage-3 feature map of both forms against each
    other; nothing may be clamped."""
    from facedetection_amd.detection import Detector
    from facedetection_amd import synthetic
    params = synthetic.synthetic_params(0)
    w = torch.from_numpy(np.asarray(params["rpn_conv_3x3_weight"], dtype=np.float32)).double()
    bias = torch.from_numpy(np.asarray(params["rpn_conv_3x3_bias"], dtype=np.float32)).double()
    wo = torch.from_numpy(np.concatenate([np.asarray(params["rpn_cls_score_weight"], dtype=np.float32).reshape(-1, 512),
                                          np.asarray(params["rpn_bbox_pred_weight"], dtype=np.float32).reshape(-1, 512)])).double()
    bo = torch.from_numpy(np.concatenate([np.asarray(params["rpn_cls_score_bias"], dtype=np.float32),
                                          np.asarray(params["rpn_bbox_pred_bias"], dtype=np.float32)])).double()
    rng = np.random.default_rng(8)
    frames = rng.integers(0, 256, size=(2, 333, 517, 3), dtype=np.uint8)
    errs, feats = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FD_WINOGRAD_SPLIT", mode)
        det = Detector("synthetic:0", 0, own_context=True, precision="f16x3", on_clamp="raise")
        det.keep_tensors(True, production=True)
        try:
            det.forward(frames)
            assert det.clamped_values() == 0
            feats[mode] = det.tensor("feat")
            feat = torch.from_numpy(feats[mode]).double().permute(0, 3, 1, 2)
            hid = torch.relu(torch.nn.functional.conv2d(feat, w, bias, padding=1))
            ref = torch.nn.functional.conv2d(hid, wo[:, :, None, None], bo).permute(0, 2, 3, 1).numpy()
            errs[mode] = float(np.abs(det.tensor("rpn_out") - ref).max()) / float(np.abs(ref).max())
        finally:
            det.close()
    assert errs["0"] <= 5e-6, errs
    assert errs["1"] <= 6 * max(errs["0"], 5e-7), errs
    scale = float(np.abs(feats["0"]).max())
    assert float(np.abs(feats["1"] - feats["0"]).max()) <= 1.5e-5 * scale
