"""CPU-side checks of the drop-in boundary: the shared library loads, exports every
symbol include/facedet.h declares, and the ctypes table covers them all.  No GPU work."""
import ctypes
import os

import numpy as np
import pytest

from facedetection_amd import _lib


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m facedetection_amd._build`"


def test_every_declared_symbol_is_exported_and_bound():
    declared = _lib.declared_symbols()
    assert len(declared) >= 12
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), "libfacedet.so does not export %s" % name
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    _lib.lib()      # sets argtypes for all of them; raises on a missing symbol


def test_version_and_error_string():
    L = _lib.lib()
    assert L.fd_version() >= 100
    assert isinstance(L.fd_last_error(), (bytes, type(None)))


def test_anchor_table_host_helper(golden):
    from facedetection_amd.symbol.processing import generate_anchors
    g = golden("anchors")
    np.testing.assert_array_equal(generate_anchors(16.0, [0.5, 1, 2], np.array([4.0, 8.0, 16.0, 32.0])),
                                  g["anchors_16_r3_s4"])
    np.testing.assert_array_equal(generate_anchors(16, [0.5, 1, 2], 2 ** np.arange(3, 6)), g["anchors_16_r3_s3"])


def test_bad_arguments_return_codes():
    L = _lib.lib()
    out = (ctypes.c_double * 4)()
    rc = L.fd_generate_anchors(16.0, None, 0, None, 0, out)
    assert rc == -1 and b"fd_generate_anchors" in L.fd_last_error()


def test_proposal_prop_protocol():
    from facedetection_amd.symbol.proposal import ProposalProp
    prop = ProposalProp("16", "(4, 8, 16, 32)", "(0.5, 1, 2)")
    assert prop.list_arguments() == ["cls_prob", "bbox_pred", "im_info"]
    assert prop.list_outputs() == ["output"]
    ins, outs = prop.infer_shape([(1, 24, 30, 40), (1, 48, 30, 40), (1, 3)])
    assert outs == [(10, 5)] and ins[2] == (1, 3)
    with pytest.raises(ValueError):
        prop.infer_shape([(2, 24, 30, 40), (2, 48, 30, 40), (2, 3)])
