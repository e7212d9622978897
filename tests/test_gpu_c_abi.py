"""The C ABI used from plain C (no Python, no torch): tests/c_abi_client.c is compiled with gcc against
include/gprc_native.h, linked only to libgprc_native.so, and run on the MI355X.  This is the shape of the
reference-side `.Call` shim."""
import os
import subprocess

import pytest

from conftest import ROOT
from gprc_amd import _native as nat

pytestmark = pytest.mark.gpu


def test_plain_c_client(tmp_path):
    exe = str(tmp_path / "c_abi_client")
    libdir = os.path.dirname(nat.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_client.c"), "-o", exe, "-L", libdir, "-lgprc_native", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
