"""The library's host code (tree builder, loaders, solve) under AddressSanitizer + UBSan
(scripts/sanitize/): sizes around every boundary, degenerate clouds, malformed files, singular
covariances.  GPU sanitizers are not available on this pool; this covers the host side only."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_code_is_clean_under_asan_ubsan():
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize", "run.sh")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "host sanitize driver: clean" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
