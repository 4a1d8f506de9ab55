"""The C-ABI library loads without a GPU and exports every symbol include/pdftable_hip.h declares."""
import ctypes
import os
import re

import pytest

from pdf_table_amd import lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from pdf_table_amd.build import build
    return build(verbose=False)


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(REPO, "include", "pdftable_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    lib = ctypes.CDLL(built)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_python_binding_covers_header(built):
    L.load()
    hdr = open(os.path.join(REPO, "include", "pdftable_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.EXPORTS.keys())


def test_plan_and_error_reporting(built):
    lib = L.load()
    nh, nw = ctypes.c_int(), ctypes.c_int()
    assert lib.pt_det_plan(1024, 1024, L.PT_DET_PRE_DB_PP, ctypes.byref(nh), ctypes.byref(nw)) == 0
    assert (nh.value, nw.value) == (960, 960)
    assert lib.pt_det_plan(640, 640, L.PT_DET_PRE_DB_PP, ctypes.byref(nh), ctypes.byref(nw)) == 0
    assert (nh.value, nw.value) == (640, 640)
    assert lib.pt_det_plan(1024, 1024, L.PT_DET_PRE_DB_TORCH, ctypes.byref(nh), ctypes.byref(nw)) == 0
    assert (nh.value, nw.value) == (736, 736)
    assert lib.pt_det_plan(100, 100, 77, ctypes.byref(nh), ctypes.byref(nw)) != 0
    assert b"flavour" in lib.pt_last_error()
    assert lib.pt_abi_version() == L.EXPECTED_ABI


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pdf_table_amd.engine import HipEngine
    with pytest.raises(L.PtError):
        HipEngine(0)
