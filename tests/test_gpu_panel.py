"""The row-panel kernel of the K = 128 1x1 layers (csrc/fd_conv_panel.hip; reference layer:
code/detection/symbol/faster_rcnn.py:45-53, conv3 + shortcut add of a bottleneck unit) against the general
direct-to-LDS kernel -- bit for bit, since both accumulate every output element in the same order -- and against
a float64 convolution, through the C ABI (fd_conv2d)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, H, W, Cout, stride, bias, relu, residual)
CASES = [
    (16, 135, 240, 512, 1, False, False, True),     # the benchmark's layer: last convolution of a 128-channel unit
    (3, 135, 240, 512, 1, False, False, True),      # 1 519 panels
    (9, 135, 239, 256, 1, True, True, True),        # ragged last panel, bias + ReLU before the residual
    (8, 270, 480, 256, 2, True, True, False),       # stride 2 (rows gathered), no residual
    (1, 135, 240, 512, 1, False, False, True),      # a single 1080p frame: 507 panels, one round of workgroups
    (1, 68, 120, 512, 1, False, False, True),       # too few panels: stays on the general kernel
]


def _run(x, w, st, b, relu, r, panel):
    from facedetection_amd import ops, profiling
    old = os.environ.get("FD_F32_PANEL")
    os.environ["FD_F32_PANEL"] = "1" if panel else "0"
    try:
        profiling.begin()
        y = ops.conv2d(x, w, stride=st, bias=b, relu=relu, residual=r)
        torch.cuda.synchronize()
        kernels = [q["kernel"] for q in profiling.end() if q["kernel"].startswith("conv_")]
    finally:
        if old is None:
            del os.environ["FD_F32_PANEL"]
        else:
            os.environ["FD_F32_PANEL"] = old
    return y, kernels


@pytest.mark.parametrize("case", CASES)
def test_panel_kernel_equals_the_general_kernel_and_float64(case):
    B, H, W, Cout, st, bias, relu, resid = case
    g = torch.Generator(device="cuda").manual_seed(H * W + Cout)
    x = torch.randn((B, H, W, 128), generator=g, device="cuda").clamp_(min=0)
    w = torch.randn((Cout, 128, 1, 1), generator=g, device="cuda") * float(np.sqrt(2.0 / 128))
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    b = torch.randn((Cout,), generator=g, device="cuda") if bias else None
    r = torch.randn((B, Ho, Wo, Cout), generator=g, device="cuda") if resid else None
    y1, k1 = _run(x, w, st, b, relu, r, True)
    y0, k0 = _run(x, w, st, b, relu, r, False)
    panels = (B * Ho * Wo + 63) // 64
    assert k0 and not any("panel" in k for k in k0)
    assert any("panel" in k for k in k1) == (panels >= 448), (k1, panels)
    assert torch.equal(y1, y0)
    # float64 on a few hundred rows spread over the tensor, the last (ragged) panel included
    M = B * Ho * Wo
    rows = np.unique(np.concatenate([np.arange(0, M, max(1, M // 300)), np.arange(max(0, M - 70), M)]))
    xs = x[:, ::st, ::st, :].reshape(M, 128)[rows].double()
    ref = xs @ w.reshape(Cout, 128).double().t()
    if bias:
        ref = ref + b.double()
    if relu:
        ref = ref.clamp(min=0)
    if resid:
        ref = ref + r.reshape(M, Cout)[rows].double()
    err = (y1.reshape(M, Cout)[rows].double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
