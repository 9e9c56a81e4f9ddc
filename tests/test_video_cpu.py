"""Host logic of the video front-end (no GPU)."""
import numpy as np
import pytest

from facedetection_amd import video
from oracle import video as ov


@pytest.mark.parametrize("seed,p,buf", [(0, 0.05, 50), (1, 0.5, 50), (2, 0.01, 8), (3, 0.2, 3), (4, 0.0, 5), (5, 1.0, 5)])
def test_face_frame_filter_vs_oracle(seed, p, buf):
    rng = np.random.default_rng(seed)
    has = (rng.uniform(size=600) < p).tolist()
    filt = video.FaceFrameFilter(buf)
    mine = []
    for i, h in enumerate(has):
        mine += filt.push(i, h)
    assert mine == ov.written_frames(has, buf)


def test_face_frame_filter_hand_worked():
    # the 3-frame look-back is flushed when the face shows up at frame 5
    has = [False] * 5 + [True] + [False] * 2
    filt = video.FaceFrameFilter(4)
    out = []
    for i, h in enumerate(has):
        out.append(filt.push(i, h))
    assert out[5] == [2, 3, 4, 5] and all(o == [] for o in out[:5])


def test_ppm_round_trip_and_sources(tmp_path):
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (5, 12, 20, 3), dtype=np.uint8)
    video.write_ppm(str(tmp_path / "a.ppm"), frames[0])
    np.testing.assert_array_equal(video.read_ppm(str(tmp_path / "a.ppm")), frames[0])
    (tmp_path / "c.ppm").write_bytes(b"P6\n# a comment\n20 12\n255\n" + frames[1].tobytes())
    np.testing.assert_array_equal(video.read_ppm(str(tmp_path / "c.ppm")), frames[1])
    np.save(str(tmp_path / "b.npy"), frames)
    got = list(video.iter_frames(str(tmp_path / "b.npy")))
    assert len(got) == 5 and all((g == f).all() for g, f in zip(got, frames))
    frames.tofile(str(tmp_path / "v.raw"))
    got = list(video.iter_frames(str(tmp_path / "v.raw"), raw_size=(20, 12)))
    assert len(got) == 5 and (got[4] == frames[4]).all()
    with pytest.raises(ValueError):
        list(video.iter_frames(str(tmp_path / "v.raw")))
    # directory: a.ppm, b.npy (5 frames), c.ppm in name order
    (tmp_path / "v.raw").unlink()
    assert len(list(video.iter_frames(str(tmp_path)))) == 7


def test_batches_split_on_shape_change():
    a, b = np.zeros((4, 6, 3), np.uint8), np.zeros((5, 6, 3), np.uint8)
    groups = list(video.batches([a, a, a, b, b, a], 2))
    assert [len(g) for g in groups] == [2, 1, 2, 1]


def test_cli_flags_match_reference_defaults():
    a = video.parse_args(["--frames", "x.npy"])
    assert (a.thresh, a.nms_thresh, a.min_size, a.scale, a.max_scale, a.gpu, a.epoch) == (0.8, 0.3, 24, 600, 1000, 0, 0)
    assert a.prefix == "mxnet-face-fr50"
