#!/usr/bin/env python3
"""Error of one convolution against float64 for the float32 kernels and (FD_BF16X3=1 in the environment) the
bf16x3 kernels: python tools/probes/b3_accuracy.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facedetection_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
for (b, h, w, cin, cout, k, kind) in ((1, 68, 120, 512, 512, 3, "relu"), (1, 68, 120, 512, 512, 3, "normal"),
                                      (2, 68, 120, 1024, 256, 1, "relu"), (4, 135, 240, 128, 512, 1, "relu"),
                                      (1, 34, 60, 512, 128, 3, "wide")):
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    if kind == "relu":
        x = np.maximum(x, 0)
    if kind == "wide":
        x = (x * np.exp2(rng.integers(-40, 40, size=x.shape))).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double(),
                                     padding=k // 2).permute(0, 2, 3, 1).numpy()
    got = ops.conv2d(x, wt, pad=k // 2, precision="f32")
    err = got.astype(np.float64) - ref
    print("%-7s K=%5d  rms err / rms out %.3e   max err / max out %.3e   mean err / rms out %+.2e"
          % (kind, cin * k * k, np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean()), np.abs(err).max() / np.abs(ref).max(),
             err.mean() / np.sqrt((ref ** 2).mean())))
