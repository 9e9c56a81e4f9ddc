#!/usr/bin/env python3
"""Compares the raw network outputs (rois, class probabilities, box deltas) of the detector with
and without the whole-unit kernel (FD_FUSE_UNITS) on a few frame sizes, including sizes whose
stage-1 map is not a multiple of the 8 x 16 pixel tile.

    python tools/fuse_check.py [mask]      (1: identity units only, 2: projection unit only, 3: both)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facedetection_amd.detection import Detector  # noqa: E402


def main():
    torch.cuda.set_device(0)
    os.environ["FD_FUSE_UNITS"] = "0"
    ref = Detector("synthetic:0", 0, own_context=True, precision="f16x3")
    os.environ["FD_FUSE_UNITS"] = sys.argv[1] if len(sys.argv) > 1 else "3"
    new = Detector("synthetic:0", 0, own_context=True, precision="f16x3")
    rng = np.random.default_rng(0)
    worst = 0.0
    for (b, h, w) in [(1, 1080, 1920), (2, 333, 517), (3, 64, 96), (1, 37, 41), (2, 600, 1000)]:
        frames = rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)
        a = ref.forward(frames)
        c = new.forward(frames)
        line = []
        for name, x, y in zip(("rois", "cls", "box"), a, c):
            x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
            d = float(np.abs(x - y).max()) if x.size else 0.0
            line.append("%s %.3g (max |v| %.3g)" % (name, d, float(np.abs(x).max()) if x.size else 0.0))
            worst = max(worst, d if name != "rois" else 0.0)
        print("%dx%dx%d: " % (b, h, w) + ", ".join(line), flush=True)
    print("worst cls/box difference %.3g" % worst)
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
