"""The first convolution of an identity-shortcut unit (code/detection/symbol/faster_rcnn.py:28-33: bn1, relu, conv1 on
the unit's raw input) in float32: the direct-to-LDS kernel that applies BatchNorm + ReLU to the activation fragments
in registers against the register-staged kernel (FD_F32_BN_FRAG=0) -- bit for bit, since both form the same activated
values and accumulate them in the same order -- and against float64, through the C ABI (fd_conv2d)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, H, W, Cin, Cout, bias, relu, fragment kernel expected)
CASES = [
    (16, 68, 120, 1024, 256, True, True, True),     # the benchmark's layer of the 256-channel stage
    (16, 135, 240, 512, 128, True, True, True),     # ... of the 128-channel stage
    (4, 270, 480, 256, 64, True, True, True),       # ... of the 64-channel stage (three workgroups per CU)
    (3, 135, 239, 512, 128, False, False, True),    # ragged last tile, no bias
    (7, 67, 119, 1024, 256, True, True, True),      # ragged
    (1, 68, 120, 1024, 256, True, True, True),      # a single frame: 64 x 64 tiles
    (1, 135, 240, 512, 128, True, True, True),      # ... 128 x 64 tiles, two workgroups per CU
    (1, 270, 480, 256, 64, True, True, False),      # ... 64 output channels on 64 x 64 tiles: register-staged kernel
    (10, 7, 7, 2048, 512, True, True, True),        # the ROI head's layer (K unsplit through this entry point)
    (300, 14, 14, 2048, 512, True, True, True),     # the ROI head's layer with many rois: 128 x 128 tiles, K = 2048
    (2, 14, 14, 4096, 512, True, True, False),      # more channels than the fragment kernel keeps in LDS
]


def _run(x, w, b, sc, sh, relu, frag):
    from facedetection_amd import ops, profiling
    old = os.environ.get("FD_F32_BN_FRAG")
    os.environ["FD_F32_BN_FRAG"] = "1" if frag else "0"
    try:
        profiling.begin()
        y = ops.conv2d(x, w, bias=b, pre_scale=sc, pre_shift=sh, relu=relu)
        torch.cuda.synchronize()
        kernels = [q["kernel"] for q in profiling.end() if q["kernel"].startswith("conv_")]
    finally:
        if old is None:
            del os.environ["FD_F32_BN_FRAG"]
        else:
            os.environ["FD_F32_BN_FRAG"] = old
    return y, kernels


@pytest.mark.parametrize("case", CASES)
def test_bn_on_fragments_equals_the_register_staged_kernel_and_float64(case):
    B, H, W, Cin, Cout, bias, relu, expect = case
    g = torch.Generator(device="cuda").manual_seed(H * W + Cin + Cout)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda") * 3.0
    w = torch.randn((Cout, Cin, 1, 1), generator=g, device="cuda") * float(np.sqrt(2.0 / Cin))
    sc = torch.rand((Cin,), generator=g, device="cuda") + 0.5
    sh = torch.randn((Cin,), generator=g, device="cuda")
    b = torch.randn((Cout,), generator=g, device="cuda") if bias else None
    y1, k1 = _run(x, w, b, sc, sh, relu, True)
    y0, k0 = _run(x, w, b, sc, sh, relu, False)
    assert k0 and all(k.endswith("_pre") for k in k0), k0
    assert (k1 and all(k.endswith("_bn") for k in k1)) == expect, k1
    assert torch.equal(y1, y0)
    M = B * H * W
    rows = np.unique(np.concatenate([np.arange(0, M, max(1, M // 300)), np.arange(max(0, M - 130), M)]))
    xs = (x.reshape(M, Cin)[rows].double() * sc.double() + sh.double()).clamp(min=0)
    ref = xs @ w.reshape(Cout, Cin).double().t()
    if bias:
        ref = ref + b.double()
    if relu:
        ref = ref.clamp(min=0)
    err = (y1.reshape(M, Cout)[rows].double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
