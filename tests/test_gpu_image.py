"""GPU resize against the NumPy restatement of OpenCV's 8-bit INTER_LINEAR (bit-exact: it is
integer arithmetic).  The oracle itself is 'parity unpinned' -- see oracle/resize.py."""
import os

import numpy as np
import pytest

from oracle import resize as orz

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D():
    from facedetection_amd import detection
    return detection


@pytest.mark.parametrize("hw", [(480, 640), (640, 480), (1080, 1920), (333, 517), (100, 1000)])
def test_resize_rule_and_pixels_bit_exact(D, hw):
    from facedetection_amd.synthetic import synthetic_image
    img = synthetic_image(hw[0], hw[1], seed=hw[0])
    out, scale = D.resize(img, 600, 1000)
    ref, ref_scale = orz.resize(img, 600, 1000)
    assert scale == ref_scale
    assert out.dtype == np.uint8 and out.shape == ref.shape
    np.testing.assert_array_equal(out, ref)


def test_resize_known_sizes(D):
    """640x480 -> 800x600 (scale 1.25); 1920x1080 -> 1000x562 (562.5 rounds to even)."""
    from facedetection_amd.synthetic import synthetic_image
    out, s = D.resize(synthetic_image(480, 640, seed=1), 600, 1000)
    assert out.shape == (600, 800, 3) and s == 1.25
    out, s = D.resize(synthetic_image(1080, 1920, seed=1), 600, 1000)
    assert out.shape == (562, 1000, 3) and s == 1000.0 / 1920.0


@pytest.mark.parametrize("fx,fy", [(0.5, 0.5), (2.0, 2.0), (0.3, 0.7), (1.0, 1.0), (1.7, 0.9)])
def test_resize_arbitrary_factors(fx, fy):
    from facedetection_amd import ops
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (121, 186, 3), dtype=np.uint8)
    np.testing.assert_array_equal(ops.resize_bilinear(img, fx, fy), orz.resize_linear_u8(img, fx, fy))


@pytest.mark.parametrize("shape,fx,fy", [((2160, 3840, 3), 1000.0 / 3840, 1000.0 / 3840),   # 4K -> 562 x 1000
                                         ((1080, 1920, 3), 0.5, 0.5),                          # exact 2x: 2x2 average
                                         ((217, 333, 3), 0.77, 0.41),                          # rows of 256 and 769 bytes: unaligned everywhere
                                         ((130, 1203, 3), 0.2, 1.0),                           # strong horizontal reduction
                                         ((97, 131, 3), 3.1, 2.3)])                            # enlargement
def test_resize_lds_kernel_shapes(shape, fx, fy):
    """Sizes chosen for the LDS-staged 3-channel kernel: tiles cut by the image edge, source rows
    whose 16-byte alignment differs from row to row, destination rows that are not a multiple of
    4 bytes, reductions and enlargements."""
    from facedetection_amd import ops
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    np.testing.assert_array_equal(ops.resize_bilinear(img, fx, fy), orz.resize_linear_u8(img, fx, fy))


# FD_FUZZ_SEEDS widens the sweep for a one-off run (default 6 seeds x 4 cases)
@pytest.mark.parametrize("seed", range(int(os.environ.get("FD_FUZZ_SEEDS", "6"))))
def test_resize_random_shapes_and_factors(seed):
    """Random frame sizes, batch sizes and factors, drawn so that both tile shapes of the LDS-staged kernel
    (256 x 4 for destinations of 512 pixels or more at reductions below 3, 128 x 8 otherwise) and the
    per-pixel fallback (reductions too strong for the staging buffer) are visited, tiles cut by both edges included."""
    from facedetection_amd import ops
    rng = np.random.default_rng(100 + seed)
    for _ in range(4):
        h, w = int(rng.integers(33, 700)), int(rng.integers(300, 2300))
        fx = float(rng.choice([rng.uniform(0.12, 0.34), rng.uniform(0.34, 1.0), rng.uniform(1.0, 1.9)]))
        fy = float(rng.uniform(0.2, 1.6))
        b = int(rng.integers(1, 4))
        batch = rng.integers(0, 256, (b, h, w, 3), dtype=np.uint8)
        out = ops.resize_bilinear(batch, fx, fy)
        for i in range(b):
            np.testing.assert_array_equal(out[i], orz.resize_linear_u8(batch[i], fx, fy),
                                          err_msg="%d x %d x %d, fx %.4f fy %.4f" % (b, h, w, fx, fy))


def test_resize_of_a_batch_slice_with_channel_swap():
    """A batch that starts at an address that is not 16-byte aligned (a slice of a larger tensor
    offset by one row of an odd-sized image), with the BGR -> RGB swap of the command-line loop."""
    import ctypes as C

    import torch
    from facedetection_amd import _lib, runtime
    rng = np.random.default_rng(8)
    frames = rng.integers(0, 256, (4, 101, 203, 3), dtype=np.uint8)
    dev = torch.from_numpy(frames).cuda()
    part = dev[1:]                                             # 101 * 203 * 3 = 61509 bytes in: odd address
    assert part.data_ptr() % 16 != 0
    fx = fy = 0.6
    dh, dw = C.c_int32(0), C.c_int32(0)
    L = _lib.lib()
    _lib.check(L.fd_resize_dims(101, 203, fx, fy, C.byref(dh), C.byref(dw)))
    out = torch.empty((3, dh.value, dw.value, 3), dtype=torch.uint8, device="cuda")
    _lib.check(L.fd_resize_cvt_u8(runtime.context(0).handle, runtime.ptr(part), 3, 101, 203, 3, fx, fy, 1, runtime.ptr(out),
                                  dh.value, dw.value, runtime.stream_ptr(0)))
    got = out.cpu().numpy()
    for b in range(3):
        np.testing.assert_array_equal(got[b], orz.resize_linear_u8(frames[1 + b], fx, fy)[:, :, ::-1])


def test_resize_batch_and_gray():
    from facedetection_amd import ops
    rng = np.random.default_rng(6)
    batch = rng.integers(0, 256, (3, 90, 160, 1), dtype=np.uint8)
    out = ops.resize_bilinear(batch, 0.52, 0.52)
    for b in range(3):
        np.testing.assert_array_equal(out[b], orz.resize_linear_u8(batch[b], 0.52, 0.52))


def test_identity_scale_is_identity():
    from facedetection_amd import ops
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
    np.testing.assert_array_equal(ops.resize_bilinear(img, 1.0, 1.0), img)


def test_cli_path_resize_then_detect_matches_oracle():
    """The loop of detection.py:105-139: resize, im_info = (h, w, scale), forward, tail without rounding."""
    import torch
    from facedetection_amd.detection import Detector, resize
    from facedetection_amd.symbol.config import config
    from facedetection_amd.synthetic import synthetic_image, synthetic_params
    from oracle import boxes as ob
    from oracle import network as onet
    params = synthetic_params(0)
    det = Detector(params, 0, precision="f16x3")
    config.TEST.RPN_MIN_SIZE = 24               # the CLI default (detection.py:180)
    det.reconfigure()
    try:
        frame = synthetic_image(360, 640, seed=9)
        img, scale = resize(frame, 300, 500)
        rois, cls_prob, deltas = det.forward(img, im_scale=scale)
        cfg = dict(pre_nms_top_n=1000, post_nms_top_n=10, nms_thresh=0.7, min_size=24)
        r_rois, r_cls, r_deltas, _ = onet.forward(img, params, cfg, torch.float32, im_scale=scale)
        np.testing.assert_allclose(rois, r_rois, rtol=1e-5, atol=0.05)
        assert np.abs(cls_prob - r_cls).max() < 1e-4
        dets = ob.postprocess(rois[:, 1:], cls_prob, deltas, img.shape[:2], do_round=False)
        ref = ob.postprocess(r_rois[:, 1:], r_cls, r_deltas, img.shape[:2], do_round=False)
        assert dets.shape == ref.shape and np.abs(dets - ref).max(initial=0) < 0.5
    finally:
        config.TEST.RPN_MIN_SIZE = 50
        det.close()
