"""icp_hip_read_csv / icp_hip_read_f32 (host code, no GPU): libcpu::read_csv restated
(libcpu/libcpu/point-3d.cc:8-134).  Checked against the reference's own compiled read_csv where
oracle/_ref exists (the build container), on the reference's data files and on hand-made CSVs
that exercise the header lookup, ignored columns and the error cases; the expected values are
also written out so the test stands on its own on a box without the reference."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits
from icp_amd import capi

CASES = {
    # name: (text, expected points or None for "must fail")
    "plain": ("Points_0,Points_1,Points_2\n1,2,3\n4.5,-6e-1,7\n", [[1, 2, 3], [4.5, -0.6, 7]]),
    "extra_columns": ("Points_0,Points_1,Points_2,Points_Magnitude,Point ID\n1,2,3,9,0\n4,5,6,9,1\n",
                      [[1, 2, 3], [4, 5, 6]]),
    "reordered": ("id,Points_2,Points_0,foo,Points_1\n7,3,1,x,2\n8,6,4,y,5\n", [[1, 2, 3], [4, 5, 6]]),
    "duplicate_name_last_wins": ("Points_0,Points_0,Points_1,Points_2\n9,1,2,3\n", [[1, 2, 3]]),
    "stof_prefixes": ("Points_0,Points_1,Points_2\n  1.5abc,0x10,inf\n", [[1.5, 16.0, np.inf]]),
    "header_only": ("Points_0,Points_1,Points_2\n", []),
    "no_trailing_newline": ("Points_0,Points_1,Points_2\n1,2,3", [[1, 2, 3]]),
    "crlf_last_column_name_breaks": ("Points_0,Points_1,Points_2\r\n1,2,3\r\n", None),
    "crlf_with_extra_column": ("Points_0,Points_1,Points_2,w\r\n1,2,3,0\r\n", [[1, 2, 3]]),
    "missing_field": ("Points_0,Points_1,Z\n1,2,3\n", None),
    "short_line": ("Points_0,Points_1,Points_2\n1,2\n", None),
    "empty_line": ("Points_0,Points_1,Points_2\n1,2,3\n\n", None),
    "not_a_number": ("Points_0,Points_1,Points_2\n1,abc,3\n", None),
    "empty_file": ("", None),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_csv_cases(tmp_path, name):
    text, want = CASES[name]
    path = tmp_path / f"{name}.txt"
    path.write_bytes(text.encode())
    from oracle import Ref
    ref_pts = None
    ref_failed = None
    if Ref.available():
        try:
            ref_pts, ref_failed = Ref().read_csv(str(path)), False
        except RuntimeError:
            ref_failed = True
    if want is None:
        with pytest.raises(capi.IcpHipError):
            capi.read_csv(str(path))
        if ref_failed is not None:
            assert ref_failed, "the reference accepts this file"
    else:
        got = capi.read_csv(str(path))
        assert got.shape == (len(want), 3)
        assert (bits(got) == bits(np.array(want, np.float32).reshape(-1, 3))).all()
        if ref_failed is not None:
            assert not ref_failed and (bits(got) == bits(ref_pts)).all()


def test_missing_file():
    with pytest.raises(capi.IcpHipError) as e:
        capi.read_csv("/nonexistent/cloud.txt")
    assert "could not open file" in str(e.value)


def test_reference_data_files(ref):
    files = sorted(glob.glob("/root/reference/data/*/*.txt"))
    if not files:
        pytest.skip("the reference's data files exist only in the build container")
    for f in files:
        assert (bits(capi.read_csv(f)) == bits(ref.read_csv(f))).all(), f


def test_golden_clouds_are_the_csv_files():
    """The .npy fixtures under tests/golden/data are read_csv outputs of the reference's files."""
    for f in sorted(glob.glob("/root/reference/data/*/*.txt")):
        npy = os.path.join(GOLDEN, "data", os.path.splitext(os.path.basename(f))[0] + ".npy")
        if os.path.exists(npy):
            assert (bits(capi.read_csv(f)) == bits(np.load(npy))).all(), f


def test_raw_float32(tmp_path):
    pts = np.random.default_rng(3).standard_normal((1000, 3)).astype(np.float32)
    p = tmp_path / "c.f32"
    pts.tofile(p)
    assert (bits(capi.read_f32(str(p))) == bits(pts)).all()
    (tmp_path / "bad.f32").write_bytes(b"\0" * 13)
    with pytest.raises(capi.IcpHipError):
        capi.read_f32(str(tmp_path / "bad.f32"))
    with pytest.raises(capi.IcpHipError):
        capi.read_f32(str(tmp_path / "absent.f32"))
