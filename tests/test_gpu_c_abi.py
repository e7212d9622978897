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


def _build_client(tmp_path, name):
    exe = str(tmp_path / name)
    libdir = os.path.dirname(nat.LIB_PATH)
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", name + ".c"), "-o", exe, "-L", libdir, "-lgprc_native", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def run_mgpu_client(tmp_path, n, d, ns, variants, timeout=600):
    """tests/c_abi_mgpu8.c: G virtual ranks on device 0 through gprc_mgpu_*, bitwise against gprc_gpr_fit / gprc_gpr_predict.
    Returns the JSON records it printed (one per variant, the single-rank line first)."""
    import json
    exe = _build_client(tmp_path, "c_abi_mgpu8")
    out = subprocess.run([exe, str(n), str(d), str(ns)] + list(variants), capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all variants bit-identical" in out.stdout
    return [json.loads(line) for line in out.stdout.splitlines() if line.startswith("{")]


def test_eight_virtual_ranks_small(tmp_path):
    """The 8-owner block-cyclic sweep at sizes where some ranks own nothing (6 panels) and where every rank owns two or
    three (18 panels): look-ahead, no look-ahead, scatter + all-gather exchange, calibration at creation -- all bit-identical to
    the single-rank entry points (R/GPRclass.R:127-170)."""
    recs = run_mgpu_client(tmp_path, 2900, 3, 777, ["8:0", "8:2", "8:4", "8:6", "5:12", "3:4", "2:4"])
    assert [r["ranks"] for r in recs[1:]] == [8, 8, 8, 8, 5, 3, 2]
    assert recs[1]["panels"] == 6 and recs[1]["exchange_mode"] == 0 and recs[3]["exchange_mode"] == 2
    recs = run_mgpu_client(tmp_path, 9100, 2, 1500, ["8:0", "8:4", "8:2"])
    look, sag, nola = recs[1:]
    assert look["panels"] == 18 and look["lookahead_updates"] == 17 and nola["lookahead_updates"] == 0
    # rooted: every panel leaves its owner 7 times (+ 7 copies of its inverses); scattered: 7 pieces + 7 x 7 pulls for the panels
    # of >= 1 MiB, and in both forms every rank receives every byte of the factor exactly once
    assert look["exchange_ops"] == 18 * 14
    assert sag["exchange_ops"] > look["exchange_ops"] and sag["gb_in_per_rank"] == look["gb_in_per_rank"]
