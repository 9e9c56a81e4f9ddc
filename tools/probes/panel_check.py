#!/usr/bin/env python3
"""The row-panel kernel of the short-K 1x1 layers against the general direct-to-LDS kernel: bit-equal outputs and
launch times, through fd_conv2d.  Each setting of FD_F32_PANEL runs in a process of its own (the library reads it
once):  python tools/probes/panel_check.py"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# (B, H, W, Cin, Cout, stride, bias, relu, residual)
SHAPES = [
    (5, 135, 240, 128, 512, 1, False, False, True),      # last convolution of a 128-channel unit
    (16, 135, 240, 128, 512, 1, False, False, True),     # ... at the benchmark's micro-batch
    (8, 270, 480, 64, 256, 1, False, False, False),      # projection shortcut of the 64-channel stage
    (8, 270, 480, 128, 256, 2, True, True, False),       # stride 2, bias + ReLU
    (9, 135, 239, 128, 256, 1, True, True, True),        # ragged last panel
    (8, 270, 479, 64, 128, 1, False, True, True),        # ragged last panel, K = 64
]


def child(out_dir):
    import torch
    from facedetection_amd import ops, profiling
    res = []
    for i, (B, H, W, Cin, Cout, st, bias, relu, resid) in enumerate(SHAPES):
        g = torch.Generator(device="cuda").manual_seed(i)
        x = torch.randn((B, H, W, Cin), generator=g, device="cuda").clamp_(min=0)
        w = torch.randn((Cout, Cin, 1, 1), generator=g, device="cuda") * float(np.sqrt(2.0 / Cin))
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        b = torch.randn((Cout,), generator=g, device="cuda") if bias else None
        r = torch.randn((B, Ho, Wo, Cout), generator=g, device="cuda") if resid else None
        y = ops.conv2d(x, w, stride=st, bias=b, relu=relu, residual=r)
        torch.cuda.synchronize()
        profiling.begin()
        for _ in range(5):
            y = ops.conv2d(x, w, stride=st, bias=b, relu=relu, residual=r)
        torch.cuda.synchronize()
        rows = [q for q in profiling.end() if q["kernel"].startswith("conv_")]
        np.save(os.path.join(out_dir, "y%d.npy" % i), y[:, ::7].cpu().numpy())
        res.append({"kernel": rows[0]["kernel"], "us": round(rows[0]["ms"] / rows[0]["launches"] * 1e3, 1),
                    "tflops": round(rows[0]["flops"] / rows[0]["ms"] / 1e9, 1), "sum": float(y.double().sum().item())})
        del x, w, y, r
    print(json.dumps(res))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    outs = {}
    with tempfile.TemporaryDirectory() as tmp:
        for flag in ("0", "1"):
            d = os.path.join(tmp, flag)
            os.makedirs(d)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", d], env=dict(os.environ, FD_F32_PANEL=flag),
                               capture_output=True, text=True, timeout=600)
            if p.returncode != 0:
                print(p.stdout[-2000:], p.stderr[-4000:])
                sys.exit(1)
            outs[flag] = json.loads([l for l in p.stdout.splitlines() if l.startswith("[")][-1])
        ok = True
        for i, s in enumerate(SHAPES):
            a, b = np.load(os.path.join(tmp, "0", "y%d.npy" % i)), np.load(os.path.join(tmp, "1", "y%d.npy" % i))
            same = bool(np.array_equal(a, b)) and outs["0"][i]["sum"] == outs["1"][i]["sum"]
            ok = ok and same
            print(s, "bit-equal" if same else "DIFFERENT (max %g)" % np.abs(a - b).max(), outs["0"][i], "->", outs["1"][i])
    print("all bit-equal" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
