"""Gadgets::divmod_u256 (halo2_vectordb_amd/csrc/gadgets.hpp: the BigUint div_mod_floor behind qdiv and div_mod_var,
/root/reference/src/gadget/fixed_point.rs:631-656) is host + device code: its host build is held to Python's integers here — every pair
of operand bit lengths from 1 to 256, equal operands, all-ones, powers of two, a zero dividend (tools/divtest.hip prints the cases).
The device build of the same source is held to the oracle by every distance / k-means parity test on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_long_division_against_python_integers(tmp_path):
    exe = str(tmp_path / "divtest")
    subprocess.check_call([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-o", exe, os.path.join(ROOT, "tools", "divtest.hip")],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) > 6000
    for line in out:
        a, b, q, r = (int(x, 16) for x in line.split())
        if b:
            assert (q, r) == (a // b, a % b), line
