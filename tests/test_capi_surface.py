"""CPU suite, part 3: the C-ABI library loads and exports every symbol include/icp_hip.h
declares; device entry points fail loudly (no fallback) when there is no GPU."""
import os
import re

import pytest

from conftest import ROOT, has_gpu
from icp_amd import capi


def declared_functions():
    text = open(os.path.join(ROOT, "include", "icp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(icp_hip_[a-z0-9_]+)\s*\(", text))
    names -= set(re.findall(r"\(\*(icp_hip_[a-z0-9_]+)\)", text))  # function-pointer typedefs
    return names


def test_every_declared_symbol_is_exported_and_bound():
    lib = capi.load()
    declared = declared_functions()
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_every_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "icp_hip.h")).read()
    for cite in ("libgpu/libgpu/icp.hh:19-21", "vp-tree.cc:54-123", "vp-tree.cu:58-112",
                 "vp-tree.cu:181-261", "icp.cu:144-158", "svd.cc:7-28", "matrix.cu:95-103"):
        assert cite in text, cite


def test_no_product_code_touches_the_oracle():
    """The oracle is test infrastructure: nothing under icp_amd/ or include/ may name it."""
    for base in ("icp_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cc", ".hh", ".h", ".hip", "Makefile")):
                    src = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "oracle" not in src.lower() or f == "capi.py" and False, os.path.join(dirpath, f)


def test_device_entry_points_fail_loudly_without_gpu():
    if has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.IcpHipError) as e:
        capi.Context()
    assert e.value.status == 3 and "HIP device" in str(e.value)
    import icp_amd
    from icp_amd import synth
    m = synth.cloud(1000)
    with pytest.raises(capi.IcpHipError):
        icp_amd.icp(m, m, 5, 1e-5, 32)
