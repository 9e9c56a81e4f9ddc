"""MXNet .params container: round trip through every record layout the reader accepts.
(No real checkpoint is available offline; the layouts are restated from the MXNet sources.)"""
import numpy as np
import pytest

from facedetection_amd import checkpoint as ck


LAYOUTS = [dict(version=0), dict(version=1), dict(version=1, v1_dims="I"), dict(version=2), dict(version=3)]


def _hand_built_v1_record(shape, values):
    """One V1 record assembled byte by byte, independently of checkpoint.save_params:
    magic, uint32 ndim, int64 dims, context (cpu, 0), dtype flag 0 (float32), payload."""
    import struct
    b = struct.pack("<II", 0xF993FAC8, len(shape))
    for d in shape:
        b += struct.pack("<q", d)
    b += struct.pack("<iii", 1, 0, 0)
    return b + np.asarray(values, "<f4").tobytes()


def test_hand_built_v1_file_with_int64_dims(tmp_path):
    import struct
    a0 = np.arange(6, dtype=np.float32).reshape(2, 3)
    a1 = np.arange(4, dtype=np.float32) - 1.5
    blob = struct.pack("<QQQ", 0x112, 0, 2) + _hand_built_v1_record((2, 3), a0) + _hand_built_v1_record((4,), a1)
    blob += struct.pack("<Q", 2)
    for name in (b"arg:w", b"aux:m"):
        blob += struct.pack("<Q", len(name)) + name
    p = tmp_path / "v1.params"
    p.write_bytes(blob)
    out = ck.load_ndarray_dict(str(p))
    np.testing.assert_array_equal(out["arg:w"], a0)
    np.testing.assert_array_equal(out["aux:m"], a1)


@pytest.mark.parametrize("layout", LAYOUTS, ids=["legacy", "v1_int64", "v1_uint32", "v2", "v3"])
def test_round_trip(tmp_path, layout):
    version = layout["version"]
    rng = np.random.default_rng(version)
    arg = {"conv0_weight": rng.normal(size=(64, 3, 7, 7)).astype(np.float32),
           "cls_score_bias": rng.normal(size=(2,)).astype(np.float32),
           "fc_f64": rng.normal(size=(3, 5))}
    aux = {"bn0_moving_mean": rng.normal(size=(64,)).astype(np.float32),
           "bn0_moving_var": rng.uniform(0.5, 2, size=(64,)).astype(np.float16)}
    path = str(tmp_path / "m-0000.params")
    ck.save_params(path, arg, aux, **layout)
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


def _small_file(tmp_path, **kw):
    rng = np.random.default_rng(5)
    path = str(tmp_path / "n.params")
    ck.save_params(path, {"a": rng.normal(size=(4, 6)).astype(np.float32),
                          "b": rng.normal(size=(7,)).astype(np.float32)}, {"c": np.ones((3,), np.float32)}, **kw)
    return path


@pytest.mark.parametrize("layout", LAYOUTS, ids=["legacy", "v1_int64", "v1_uint32", "v2", "v3"])
@pytest.mark.parametrize("cut", [9, 40, 100, 160])
def test_truncated_files_are_rejected(tmp_path, layout, cut):
    path = _small_file(tmp_path, **layout)
    data = open(path, "rb").read()
    assert cut < len(data)
    open(path, "wb").write(data[:len(data) - cut])
    with pytest.raises(ValueError):
        ck.load_ndarray_dict(path)


@pytest.mark.parametrize("layout", LAYOUTS, ids=["legacy", "v1_int64", "v1_uint32", "v2", "v3"])
def test_name_count_mismatch_is_rejected(tmp_path, layout):
    path = _small_file(tmp_path, n_names=2, **layout)
    with pytest.raises(ValueError):
        ck.load_ndarray_dict(path)


def test_wrong_record_magic_and_dtype_are_rejected(tmp_path):
    import struct
    path = _small_file(tmp_path, version=2)
    data = bytearray(open(path, "rb").read())
    bad = bytearray(data)
    struct.pack_into("<I", bad, 24, 0xF993FACB)        # first record's magic: not a known version
    open(path, "wb").write(bad)
    with pytest.raises(ValueError):
        ck.load_ndarray_dict(path)
    bad = bytearray(data)
    struct.pack_into("<i", bad, 24 + 8 + 4 + 16 + 8, 9)  # magic+stype, ndim, 2 dims, context -> dtype flag
    open(path, "wb").write(bad)
    with pytest.raises(ValueError, match="dtype"):
        ck.load_ndarray_dict(path)


def test_sparse_records_are_refused(tmp_path):
    import struct
    path = _small_file(tmp_path, version=2)
    data = bytearray(open(path, "rb").read())
    struct.pack_into("<i", data, 28, 1)                # storage type: row-sparse
    open(path, "wb").write(data)
    with pytest.raises(ValueError, match="sparse"):
        ck.load_ndarray_dict(path)


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
