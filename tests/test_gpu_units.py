"""The whole-unit kernels of the 64-channel stage (a residual unit's three convolutions, and for the
first unit also its shortcut convolution, in one launch) against the same unit run as separate
convolution kernels.  Both paths compute the same float32-grade products; they differ in the order
in which the three partial products of a split multiply are added, so tensors agree to a few ulps
rather than bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# frame sizes whose stage-1 map (a quarter of the frame) is not a multiple of the 8 x 16 pixel tile,
# is smaller than one tile, and spans many tiles
SIZES = [(1, 37, 41), (3, 64, 96), (2, 333, 517), (1, 270, 480)]


def _detector(monkeypatch, mask, keep=False):
    from facedetection_amd.detection import Detector
    monkeypatch.setenv("FD_FUSE_UNITS", str(mask))
    det = Detector("synthetic:0", 0, own_context=True, precision="f16x3")
    if keep:
        det.keep_tensors(True)
    return det


@pytest.mark.parametrize("mask", [1, 2, 3])
def test_fused_units_give_the_outputs_of_the_separate_kernels(monkeypatch, mask):
    ref = _detector(monkeypatch, 0)
    new = _detector(monkeypatch, mask)
    rng = np.random.default_rng(mask)
    try:
        for (b, h, w) in SIZES:
            frames = rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)
            rois_a, cls_a, box_a = ref.forward(frames)
            rois_b, cls_b, box_b = new.forward(frames)
            np.testing.assert_allclose(rois_b, rois_a, atol=0.01)      # pixels
            np.testing.assert_allclose(cls_b, cls_a, atol=1e-5)
            np.testing.assert_allclose(box_b, box_a, atol=1e-5)
    finally:
        ref.close()
        new.close()


def test_fused_unit_tensors_match_elementwise(monkeypatch):
    """Unit outputs as tensors (mask bit 2 keeps the fused kernels on while tensors are recorded)."""
    ref = _detector(monkeypatch, 0, keep=True)
    new = _detector(monkeypatch, 7, keep=True)
    rng = np.random.default_rng(7)
    try:
        for (b, h, w) in SIZES:
            frames = rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)
            ref.forward(frames)
            new.forward(frames)
            for name in ("stage1_unit1", "stage1_unit2", "stage1_unit3"):
                want, got = ref.tensor(name), new.tensor(name)
                assert want.shape == got.shape
                scale = float(np.abs(want).max())
                assert float(np.abs(got - want).max()) <= 2e-6 * scale, (name, (b, h, w))
    finally:
        ref.close()
        new.close()


@pytest.mark.parametrize("tail", ["winograd+unit", "winograd+shortcut", "winograd", "direct"])
def test_float32_fused_tail_matches_the_separate_float32_kernels(monkeypatch, tail):
    """precision="f32": an ordinary pass runs the 3x3 + 1x1 tail of each 64-channel unit as one kernel (the tensor
    between them stays in LDS as float32) -- with the 3x3 in Winograd F(2x2,3x3) form (fd_conv_tailw.hip: transforms
    in registers, nothing of the Winograd domain in memory; the default) or, with FD_TAIL_WINO=0, as a direct
    convolution (conv_tail_fused_kernel); a layer-by-layer pass runs two direct kernels.  Same float32 products up to
    the order of summation (and, for the Winograd form, its additions-only transforms), so unit outputs agree to
    rounding noise.  By default the Winograd tail of the first unit also holds that unit's shortcut convolution
    (faster_rcnn.py:45-53: conv3 + shortcut = one 1x1 over the 64 + 64 channels of t2 and the activated input; the
    shortcut tensor is neither written nor read) and its conv1, computed on the input patch in LDS (the whole unit is
    one launch); FD_TAIL_C1=0 keeps conv1, FD_TAIL_SC=0 the shortcut as well, a launch of its own.  The frame
    sizes give stage-1 maps smaller than one 8 x 16 tile, ragged in both directions, and many tiles large."""
    from facedetection_amd import profiling
    from facedetection_amd.detection import Detector
    monkeypatch.setenv("FD_TAIL_WINO", "0" if tail == "direct" else "1")
    monkeypatch.setenv("FD_TAIL_SC", "1" if tail in ("winograd+shortcut", "winograd+unit") else "0")
    monkeypatch.setenv("FD_TAIL_C1", "1" if tail == "winograd+unit" else "0")
    lay = Detector("synthetic:0", 0, own_context=True, precision="f32")
    new = Detector("synthetic:0", 0, own_context=True, precision="f32")
    lay.keep_tensors(True)
    new.keep_tensors(True, production=True)
    rng = np.random.default_rng(11)
    try:
        for (b, h, w) in SIZES:
            frames = rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)
            lay.forward(frames)
            profiling.begin()
            new.forward(frames)
            rows = profiling.end()
            ran = {r["kernel"] for r in rows}
            assert ("conv_tail_wino_f32" in ran) == (tail != "direct") and ("conv_tail_fused_f32" in ran) == (tail == "direct"), ran
            assert ("conv_tail_wino_sc_f32" in ran) == (tail == "winograd+shortcut"), ran
            assert ("unit_proj_wino_f32" in ran) == (tail == "winograd+unit"), ran
            tails = sum(r["launches"] for r in rows if r["kernel"].startswith(("conv_tail_", "unit_proj_wino")))
            assert tails == 3, rows
            for name in ("stage1_unit1", "stage1_unit2", "stage1_unit3"):
                want, got = lay.tensor(name), new.tensor(name)
                assert want.shape == got.shape
                scale = float(np.abs(want).max())
                assert float(np.abs(got - want).max()) <= 4e-6 * scale, (name, (b, h, w))
    finally:
        lay.close()
        new.close()


def test_paired_shortcut_and_conv1_launch_gives_the_same_bits(monkeypatch):
    """precision="f32": the first unit of the 64-channel stage runs its shortcut convolution and its conv1 -- both 1x1,
    stride 1, same input -- as ONE launch of the direct-to-LDS kernel when FD_F32_PAIR=1 (weight rows concatenated, two
    output tensors; faster_rcnn.py:33,51; off by default).  Every output element accumulates and is finished exactly as
    in two launches, so the unit's output is the same bit for bit."""
    from facedetection_amd import profiling
    from facedetection_amd.detection import Detector
    monkeypatch.setenv("FD_TAIL_SC", "0")       # (with the shortcut inside the tail there is no shortcut launch to pair)
    det = Detector("synthetic:0", 0, own_context=True, precision="f32")
    det.keep_tensors(True, production=True)
    rng = np.random.default_rng(5)
    try:
        for (b, h, w) in [(2, 1080, 1920), (3, 1000, 1500)]:      # (the form is taken from 768 row tiles on)
            frames = rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)
            got, launches = {}, {}
            for pair in ("1", "0"):
                monkeypatch.setenv("FD_F32_PAIR", pair)
                profiling.begin()
                det.forward(frames)
                rows = {r["kernel"]: r["launches"] for r in profiling.end()}
                launches[pair] = sum(n for k, n in rows.items() if k.startswith("conv_"))
                got[pair] = {name: det.tensor(name).copy() for name in ("stage1_unit1", "stage1_unit3", "feat")}
            assert launches["0"] == launches["1"] + 1, launches          # one convolution launch fewer
            for name in got["1"]:
                assert np.array_equal(got["1"][name], got["0"][name]), (name, (b, h, w))
    finally:
        det.close()


def test_shortcut_and_conv1_inside_the_tail_save_launches_and_keep_the_values(monkeypatch):
    """precision="f32", first unit of the 64-channel stage: with the shortcut convolution inside the Winograd tail a
    pass has one convolution launch fewer than with FD_TAIL_SC=0, with conv1 inside as well (default) two fewer, and
    the unit outputs agree to rounding noise (the shortcut's products join conv3's accumulators instead of being added
    as a finished tensor; conv1's values are the same products, formed in the kernel)."""
    from facedetection_amd import profiling
    from facedetection_amd.detection import Detector
    det = Detector("synthetic:0", 0, own_context=True, precision="f32")
    det.keep_tensors(True, production=True)
    rng = np.random.default_rng(6)
    try:
        for (b, h, w) in [(2, 1080, 1920), (1, 333, 517), (1, 37, 41)]:
            frames = rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)
            got, launches = {}, {}
            for form in ("unit", "shortcut", "separate"):
                monkeypatch.setenv("FD_TAIL_SC", "0" if form == "separate" else "1")
                monkeypatch.setenv("FD_TAIL_C1", "1" if form == "unit" else "0")
                profiling.begin()
                det.forward(frames)
                rows = {r["kernel"]: r["launches"] for r in profiling.end()}
                launches[form] = sum(n for k, n in rows.items() if k.startswith(("conv_", "unit_proj_wino")))
                got[form] = {name: det.tensor(name).copy() for name in ("stage1_unit1", "stage1_unit3", "feat")}
            assert launches["separate"] == launches["shortcut"] + 1 == launches["unit"] + 2, launches
            for form in ("unit", "shortcut"):
                for name in got[form]:
                    scale = float(np.abs(got["separate"][name]).max())
                    assert float(np.abs(got[form][name] - got["separate"][name]).max()) <= 4e-6 * scale, (form, name, (b, h, w))
    finally:
        det.close()


def test_projection_shortcut_inside_conv3_of_the_later_stages(monkeypatch):
    """precision="f32", first units of the 128- and 256-channel stages (stride 2; faster_rcnn.py:45-53): conv3 and the
    projection shortcut run as ONE GEMM whose rows take their channels from two tensors -- t2's row, then the
    activated input's pixel (2 ho, 2 wo) -- against the weight rows [W3 | Wsc]; FD_SC_IN_C3=0 runs the shortcut as a
    launch of its own and adds its tensor as conv3's residual.  Two launches fewer per pass, unit outputs equal to
    rounding noise; frame sizes with ragged maps and a map smaller than one tile."""
    from facedetection_amd import profiling
    from facedetection_amd.detection import Detector
    det = Detector("synthetic:0", 0, own_context=True, precision="f32")
    det.keep_tensors(True, production=True)
    rng = np.random.default_rng(8)
    try:
        for (b, h, w) in [(2, 1080, 1920), (3, 333, 517), (1, 37, 41)]:
            frames = rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)
            got, launches, names = {}, {}, {}
            for form in ("1", "0"):
                monkeypatch.setenv("FD_SC_IN_C3", form)
                profiling.begin()
                det.forward(frames)
                rows = {r["kernel"]: r["launches"] for r in profiling.end()}
                names[form] = set(rows)
                launches[form] = sum(n for k, n in rows.items() if k.startswith(("conv_", "unit_proj_wino")))
                got[form] = {name: det.tensor(name).copy() for name in ("stage2_unit1", "stage2_unit4", "stage3_unit1", "feat")}
            assert launches["0"] == launches["1"] + 2, launches
            assert any(k.endswith("_two") for k in names["1"]) and not any(k.endswith("_two") for k in names["0"]), names
            for name in got["1"]:
                scale = float(np.abs(got["0"][name]).max())
                assert float(np.abs(got["1"][name] - got["0"][name]).max()) <= 4e-6 * scale, (name, (b, h, w))
    finally:
        det.close()
