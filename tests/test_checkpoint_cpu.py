"""MXNet .params container: round trip through every record layout the reader accepts.
(No real checkpoint is available offline; the layouts are restated from the MXNet sources.)"""
import numpy as np
import pytest

from facedetection_amd import checkpoint as ck


@pytest.mark.parametrize("version", [0, 1, 2, 3])
def test_round_trip(tmp_path, version):
    rng = np.random.default_rng(version)
    arg = {"conv0_weight": rng.normal(size=(64, 3, 7, 7)).astype(np.float32),
           "cls_score_bias": rng.normal(size=(2,)).astype(np.float32),
           "fc_f64": rng.normal(size=(3, 5))}
    aux = {"bn0_moving_mean": rng.normal(size=(64,)).astype(np.float32),
           "bn0_moving_var": rng.uniform(0.5, 2, size=(64,)).astype(np.float16)}
    path = str(tmp_path / "m-0000.params")
    ck.save_params(path, arg, aux, version=version)
    raw = ck.load_ndarray_dict(path)
    assert set(raw) == {"arg:" + k for k in arg} | {"aux:" + k for k in aux}
    np.testing.assert_array_equal(raw["arg:conv0_weight"], arg["conv0_weight"])
    assert raw["arg:fc_f64"].dtype == np.float64
    merged = ck.load_params(path)
    assert set(merged) == set(arg) | set(aux)
    assert all(v.dtype == np.float32 for v in merged.values())
    np.testing.assert_array_equal(merged["bn0_moving_var"], aux["bn0_moving_var"].astype(np.float32))


def test_rejects_other_files(tmp_path):
    p = tmp_path / "x.params"
    p.write_bytes(b"\x00" * 64)
    with pytest.raises(ValueError):
        ck.load_ndarray_dict(str(p))


def test_synthetic_checkpoint_is_loadable_by_prefix(tmp_path):
    from facedetection_amd.detection import load_model
    from facedetection_amd.synthetic import bn_shapes, synthetic_params
    p = synthetic_params(3)
    aux_names = {n + s for n, _ in bn_shapes() for s in ("_moving_mean", "_moving_var")}
    arg = {k: v for k, v in p.items() if k not in aux_names}
    aux = {k: v for k, v in p.items() if k in aux_names}
    prefix = str(tmp_path / "mxnet-face-fr50")
    ck.save_params(prefix + "-0000.params", arg, aux)
    back = load_model(prefix, 0)
    assert set(back) == set(p)
    np.testing.assert_array_equal(back["stage3_unit2_conv2_weight"], p["stage3_unit2_conv2_weight"])
