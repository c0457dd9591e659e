"""The C++ host mirror (halo2_vectordb_amd/host/vectordb.hpp) driven like examples/kmeans.rs, checked
against the oracle and the f64 k-means of the reference's tests (tests/vectordb/mod.rs:31-91)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_kmeans_matches_oracle(O):
    exe = os.path.join(ROOT, "halo2_vectordb_amd", "host", "example_kmeans")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "halo2_vectordb_amd", "csrc"), "../host/example_kmeans"])
    rng = np.random.default_rng(31)
    vecs = rng.random((30, 5))  # tests/vectordb_test.rs:12-29 shape: kmeans<2,4> over 30 x 5
    txt = "30 5\n" + "\n".join(" ".join(repr(float(x)) for x in v) for v in vecs) + "\n"
    out = subprocess.run([exe, "euclidean"], input=txt, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    qv = O.quantize(vecs)
    c = O.Ctx()
    c.assign_witnesses(qv)
    cent, ind = c.kmeans("euclidean", qv, 2, 4)
    assert lines[0] == f"cells {len(c)} lookups {c.n_lookup}"
    got = np.array([[float(x) for x in ln.split()] for ln in lines[1:3]])
    assert np.array_equal(got, O.dequantize(cent))
    ids = [int(x) for x in lines[3].split()]
    cf = vecs[:2].copy()
    for _ in range(4):
        fid = np.array([int(np.argmin([np.linalg.norm(v - cc) for cc in cf])) for v in vecs])
        cf = np.array([vecs[fid == k].mean(axis=0) for k in range(2)])
    assert ids == list(fid)
    assert np.allclose(got, cf, rtol=1e-6)
