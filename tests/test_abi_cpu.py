"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/upsnet_hip.h declares
(no compute calls without a GPU); the product fails loudly when asked to compute without its HIP path."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "upsnet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(upsnet_[a-z0-9_]+)\s*\(", txt)))


def test_build_and_symbols():
    from upsnet_amd import build, _lib
    path = build.build(verbose=False)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    declared = header_symbols()
    assert len(declared) >= 20
    for s in declared:
        assert hasattr(lib, s), "header declares %s but the library does not export it" % s
    assert sorted(_lib.exported_symbols()) == declared, "ctypes binding table and header disagree"
    assert _lib.lib().upsnet_abi_version() == 1
    assert _lib.lib().upsnet_last_error() == b""


def test_host_side_argument_validation_without_gpu():
    """Entry points validate arguments before touching the device: error code + message, no crash."""
    from upsnet_amd import _lib
    lib = _lib.lib()
    assert lib.upsnet_roi_align_forward(None, None, 1.0, 1, 4, 4, 4, 7, 7, 2, None, None) != 0
    assert b"null pointer" in lib.upsnet_last_error()
    four = (ctypes.c_void_p * 4)(1, 1, 1, 1)
    dims = (ctypes.c_int * 4)(8, 8, 8, 8)
    sc = (ctypes.c_float * 4)(0.25, 0.125, 0.0625, 0.03125)
    assert lib.upsnet_fpn_roi_align_forward(None, four, dims, dims, sc, 6, ctypes.c_void_p(1), 1, None, 7, 7, 2, ctypes.c_void_p(1), None) != 0
    assert b"multiple of 4" in lib.upsnet_last_error()
    assert lib.upsnet_nms_workspace_bytes(5, 1000) > 5 * 1000 * 16 * 8
    assert lib.upsnet_mask_roi_capacity(1000, 9, 0) == 8000 and lib.upsnet_mask_roi_capacity(1000, 9, 1) == 8000
    assert lib.upsnet_mask_roi_capacity(300, 81, 1) == 8192


def test_no_cpu_fallback():
    """Non-CUDA tensors raise, exactly like the reference Functions (functions/deform_conv.py:40-41)."""
    from upsnet_amd import ops
    from upsnet_amd.operators.functions.roialign import RoIAlignFunction
    with pytest.raises(Exception):
        RoIAlignFunction(7, 7, 0.25)(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5))
    with pytest.raises(Exception):
        ops.gpu_nms(torch.zeros(3, 5), 0.5)
    with pytest.raises(Exception):
        ops.roi_align_nchw(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 7, 7, 0.25)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under upsnet_amd/ may import it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "upsnet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
