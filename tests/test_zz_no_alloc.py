"""fd_nms_f32_noalloc (greedy NMS on the library's fixed device workspace, no allocation) against
fd_nms_f32 through the C ABI: two calls in a row, the second with a smaller input, then the edges."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _boxes(n, seed):
    """Clustered boxes with distinct scores (float32 [n,5])."""
    g = np.random.default_rng(seed)
    centres = g.uniform(0, 1000, (max(1, n // 6), 2))
    c = centres[g.integers(0, len(centres), n)] + g.normal(0, 6, (n, 2))
    wh = g.uniform(12, 60, (n, 2))
    d = np.empty((n, 5), np.float32)
    d[:, 0:2] = c - wh / 2
    d[:, 2:4] = c + wh / 2
    d[:, 4] = g.permutation(n).astype(np.float32) / n
    return d


def _call(fn, ctx, dets, thresh):
    from facedetection_amd import _lib, runtime
    n = dets.shape[0]
    d = torch.from_numpy(dets).cuda() if n else torch.empty((0, 5), device="cuda")
    keep = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
    nk = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    stream = runtime.stream_ptr(0)
    args = [runtime.ptr(d), n, C.c_float(thresh), runtime.ptr(keep), runtime.ptr(nk), stream]
    rc = fn(*([ctx] + args if ctx is not None else args))
    torch.cuda.synchronize()
    return rc, keep.cpu().numpy()[:max(int(nk.item()), 0)], int(nk.item())


def test_no_alloc_nms_equals_the_allocating_entry_point():
    from facedetection_amd import _lib, runtime
    L = _lib.lib()
    ctx = runtime.context(0).handle
    with torch.cuda.device(0):
        for n, seed, thresh in ((3000, 1, 0.7), (700, 2, 0.3)):        # second call: smaller input, stale workspace
            dets = _boxes(n, seed)
            rc_a, keep_a, nk_a = _call(L.fd_nms_f32, ctx, dets, thresh)
            rc_b, keep_b, nk_b = _call(L.fd_nms_f32_noalloc, None, dets, thresh)
            assert rc_a == 0 and rc_b == 0, L.fd_last_error()
            assert nk_a == nk_b and 0 < nk_a < n
            np.testing.assert_array_equal(keep_a, keep_b)              # index work: bit-exact
        # the limits: the largest size, empty input, one box too many
        dets = _boxes(4096, 3)
        _, keep_a, nk_a = _call(L.fd_nms_f32, ctx, dets, 0.5)
        rc, keep_b, nk_b = _call(L.fd_nms_f32_noalloc, None, dets, 0.5)
        assert rc == 0 and nk_a == nk_b
        np.testing.assert_array_equal(keep_a, keep_b)
        rc, _, nk = _call(L.fd_nms_f32_noalloc, None, np.zeros((0, 5), np.float32), 0.5)
        assert rc == 0 and nk == 0
        rc, _, _ = _call(L.fd_nms_f32_noalloc, None, _boxes(4097, 4), 0.5)
        assert rc == -3 and b"4096" in L.fd_last_error()                 # FD_ERR_LIMIT, nothing launched


@pytest.mark.parametrize("tag,thresh", [("nms_n1000", 0.7), ("nms_n1000", 0.3), ("nms_n10", 0.3), ("nms_n1", 0.7)])
def test_no_alloc_nms_against_reference_vectors(tag, thresh):
    """The no-allocation entry point on its own against keep lists the reference's nms produced
    (code/detection/symbol/processing.py:114-148; tests/golden/nms.npz, written by tests/golden/make_golden.py)."""
    import os
    from facedetection_amd import _lib
    L = _lib.lib()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "nms.npz"))
    dets = np.ascontiguousarray(g[tag + "_dets"], dtype=np.float32)
    want = g["%s_t%d_keep" % (tag, round(thresh * 10))]
    with torch.cuda.device(0):
        rc, keep, nk = _call(L.fd_nms_f32_noalloc, None, dets, thresh)
    assert rc == 0, L.fd_last_error()
    assert nk == len(want)
    np.testing.assert_array_equal(keep, want)
