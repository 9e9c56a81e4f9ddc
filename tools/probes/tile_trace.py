#!/usr/bin/env python3
"""Phase timestamps of the direct-to-LDS convolution kernel, one layer alone (library built with -DFD_TILE_TRACE):
    python tools/probes/tile_trace.py B H W Cin Cout [residual=1] [kh=1]
Slots per wave: 0 entry, 1 first stage landed, 2 K loop done, 3 residual of row group 0 arrived (+ barrier),
4 group 0 stored, 5 residual of group 1 arrived, 6 group 1 stored, 7 stores drained.  100 MHz ticks."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facedetection_amd import _lib, ops  # noqa: E402

B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
res = int(sys.argv[6]) if len(sys.argv) > 6 else 1
kh = int(sys.argv[7]) if len(sys.argv) > 7 else 1
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((B, H, W, Cin), generator=g, device="cuda")
w = torch.randn((Cout, Cin, kh, kh), generator=g, device="cuda") * float(np.sqrt(2.0 / (Cin * kh * kh)))
r = torch.randn((B, H, W, Cout), generator=g, device="cuda") if res else None
for _ in range(3):
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    y = ops.conv2d(x, w, pad=kh // 2, precision="f32", residual=r)
    t1.record()
    torch.cuda.synchronize()
print("call %.3f ms (incl. weight packing)" % t0.elapsed_time(t1))
L = C.CDLL(_lib.lib()._name)
n_wg = min(1 << 16, ((B * H * W + 127) // 128) * ((Cout + 127) // 128))
buf = np.zeros((n_wg, 4, 8), np.uint64)
assert L.fd_debug_trace_read(buf.ctypes.data_as(C.c_void_p), n_wg) == 0
t = buf.astype(np.int64)
t = t[(t[:, 0, 0] > 0)]
print("workgroups traced:", len(t))
base = t[:, :, 0].min()
us = (t - base) / 100.0
d = np.diff(us, axis=2)                                     # [wg, wave, 7] phase durations
names = ["entry->first stage", "K loop", "res0+barrier", "group0 out", "res1 wait", "group1 out", "drain"]
for i, n in enumerate(names):
    v = d[:, :, i].ravel()
    print("%-20s median %6.2f us   p10 %6.2f  p90 %6.2f" % (n, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
life = us[:, :, 7].max(axis=1) - us[:, :, 0].min(axis=1)
print("workgroup lifetime median %.2f us p90 %.2f; kernel span %.1f us; sum of lifetimes / span = %.1f workgroups in flight"
      % (np.median(life), np.percentile(life, 90), us.max(), life.sum() / us.max()))
# first wave of workgroups (all start together) vs later ones
start = us[:, :, 0].min(axis=1)
order = np.argsort(start)
first = order[:512]
late = order[len(order) // 2: len(order) // 2 + 512]
for name, idx in (("first 512", first), ("middle 512", late)):
    print(name, "lifetime %.2f us; phases" % np.median(life[idx]), np.round(np.median(d[idx].reshape(-1, 7), axis=0), 2))
