#!/usr/bin/env python3
"""Runs one convolution shape a few times (for rocprofv3 counter passes).
    python tools/conv_probe.py B H W Cin Cout k stride pad [precision] [reps] [pre]
A last argument "pre" feeds the input through a per-channel scale, shift and ReLU (the form the
first convolution of an identity-shortcut unit takes)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facedetection_amd import ops  # noqa: E402

B, H, W, Cin, Cout, k, stride, pad = (int(v) for v in sys.argv[1:9])
precision = sys.argv[9] if len(sys.argv) > 9 else "f16x3"
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 5
pre = len(sys.argv) > 11 and sys.argv[11] == "pre"
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((B, H, W, Cin), generator=g, device="cuda")
w = torch.randn((Cout, Cin, k, k), generator=g, device="cuda") * float(np.sqrt(2.0 / (Cin * k * k)))
ps = torch.rand((Cin,), generator=g, device="cuda") + 0.5 if pre else None
ph = torch.randn((Cin,), generator=g, device="cuda") * 0.1 if pre else None
for _ in range(reps):
    y = ops.conv2d(x, w, stride=stride, pad=pad, precision=precision, pre_scale=ps, pre_shift=ph)
torch.cuda.synchronize()
print("done", tuple(y.shape))
