"""Host-side checks of the crop oracle on hand-worked cases (Function.py:49-70,95-113)."""
import numpy as np

from oracle import crops as oc


def test_stick_rects_hand_worked():
    # 68 px wide -> pad round(6.8) = 7; 25 px -> round(2.5) = 2 (half to even)
    r = oc.stick_rects(np.array([[10.9, 20.2, 78.9, 45.7, 0.9]]), 100, 200)
    assert r.tolist() == [[3, 18, 85, 47]]
    # clamped to the frame; negative coordinates truncate toward zero
    r = oc.stick_rects(np.array([[-5.5, -0.5, 30.0, 99.0]]), 100, 32)
    assert r.tolist() == [[0, 0, 32, 100]]


def test_resize_pad_letterbox_geometry():
    img = np.full((50, 100, 3), 200, dtype=np.uint8)            # wide: fills the width
    out = oc.resize_pad(img, (112, 112))
    assert out.shape == (112, 112, 3)
    rows = np.where(out[:, 56, 0] > 0)[0]
    assert rows.min() == 28 and rows.max() == 83 and (out[28:84] == 200).all()
    img = np.full((100, 40, 3), 90, dtype=np.uint8)             # tall: fills the height
    out = oc.resize_pad(img, (112, 112))
    cols = np.where(out[56, :, 0] > 0)[0]
    assert cols.size == 45 and cols.min() == (112 - 45) // 2
