#!/usr/bin/env python3
"""One 1x1 layer shape with and without the residual input, several times each (for a kernel trace):
    python tools/probes/c3_probe.py B H W Cin Cout [reps]
The trace shows what the residual read costs a short-K layer whose epilogue traffic is as long as its K loop."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facedetection_amd import ops  # noqa: E402

B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((B, H, W, Cin), generator=g, device="cuda")
w = torch.randn((Cout, Cin, 1, 1), generator=g, device="cuda") * float(np.sqrt(2.0 / Cin))
r = torch.randn((B, H, W, Cout), generator=g, device="cuda")
for mode in ("res", "nores"):
    for _ in range(reps):
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        y = ops.conv2d(x, w, precision="f32", residual=r if mode == "res" else None)
        t1.record()
        torch.cuda.synchronize()
    print(mode, "%.3f ms (incl. weight packing)" % t0.elapsed_time(t1))
