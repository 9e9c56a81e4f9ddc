#!/usr/bin/env python3
"""One-off robustness run on random frame sizes and batch sizes: the kernels of an ordinary pass (fused stem,
fused tails or whole-unit kernels, Winograd layers, the large tiles) against the layer-by-layer pass of the same
detector, in either precision.  Prints the worst relative difference per tensor; exits non-zero above 3e-5.

    python tools/fuzz_paths.py [cases=12] [seed=0] [precision=f32] [max_h=420] [max_w=700] [max_batch=3]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facedetection_amd.detection import Detector  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    precision = sys.argv[3] if len(sys.argv) > 3 else "f32"
    max_h = int(sys.argv[4]) if len(sys.argv) > 4 else 420
    max_w = int(sys.argv[5]) if len(sys.argv) > 5 else 700
    max_b = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    lay = Detector("synthetic:0", 0, own_context=True, precision=precision, on_clamp="raise")
    new = Detector("synthetic:0", 0, own_context=True, precision=precision, on_clamp="raise")
    lay.keep_tensors(True)
    new.keep_tensors(True, production=True)
    worst = {}
    try:
        for _ in range(cases):
            b = int(rng.integers(1, max_b + 1))
            h, w = int(rng.integers(32, max_h)), int(rng.integers(32, max_w))
            frames = rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)
            lay.forward(frames)
            new.forward(frames)
            for name in ("stage1_unit1", "stage1_unit3", "stage2_unit4", "feat", "rpn_out"):
                a, c = lay.tensor(name), new.tensor(name)
                assert a.shape == c.shape, (name, a.shape, c.shape)
                assert np.isfinite(c).all(), (name, b, h, w)
                rel = float(np.abs(a - c).max()) / max(float(np.abs(a).max()), 1e-30)
                worst[name] = max(worst.get(name, 0.0), rel)
            print("%d x %3d x %3d ok" % (b, h, w), flush=True)
    finally:
        lay.close()
        new.close()
    print(worst)
    sys.exit(0 if max(worst.values()) <= 3e-5 else 1)


if __name__ == "__main__":
    main()
