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


def test_example_prove_commits_what_the_oracle_commits(O, tmp_path):
    """the Prove arm in compiled code (host/example_prove.cpp): witness -> commit from the stream -> NTTs through the C ABI
    alone; the commitments of the first / last advice column and the first lookup column are recomputed by the oracle's MSM
    from the Lagrange columns the program dumps, over the SRS of the same tau; the cell counts are the oracle's"""
    exe = os.path.join(ROOT, "halo2_vectordb_amd", "host", "example_prove")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "halo2_vectordb_amd", "csrc"), "../host/example_prove"])
    rng = np.random.default_rng(77)
    n, dim, K, I, k, L = 12, 6, 2, 2, 11, 10
    vecs = rng.integers(0, 219, size=(n, dim)).astype(np.float64)
    txt = f"{n} {dim} {K} {I} {k} {L}\n" + "\n".join(" ".join(repr(float(x)) for x in v) for v in vecs) + "\n"
    tau, dump = 987654321987654321, tmp_path / "dump.bin"
    out = subprocess.run([exe, str(tau), str(dump)], input=txt, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    words = out.stdout.split()
    qv = O.quantize(vecs)
    c = O.Ctx()
    c.assign_witnesses(qv)
    c.kmeans("euclidean", qv, K, I, L=L)
    assert int(words[1]) == len(c) and int(words[3]) == c.n_lookup
    raw = np.fromfile(dump, dtype=np.uint64)
    n_adv, n_lk, rows = (int(x) for x in raw[:3])
    assert rows == 1 << k and n_adv == int(words[5]) and n_lk == int(words[7]) and n_adv >= 2 and n_lk >= 1
    commits = raw[3: 3 + (n_adv + n_lk) * 8].reshape(-1, 8)
    cols = raw[3 + (n_adv + n_lk) * 8:].reshape(3, rows, 4)
    _, gl = O.srs_from_tau(k, tau)
    want = O.msm_batch(cols, gl)
    assert np.array_equal(commits[[0, n_adv - 1, n_adv]], want)
    # the first advice column starts with the assigned vectors, the lookup column with range-checked cells
    assert np.array_equal(cols[0, : n * dim], qv.reshape(-1, 4))
    assert np.array_equal(cols[2, :8], c.lookup()[:8])


def test_example_mock_is_the_mock_arm_in_compiled_code(O):
    """host/example_mock.cpp: the reference's Mock arm (src/scaffold/mod.rs:263-266) for the distance examples through the C ABI
    alone — witness with keygen flags, device MockProver — on data/euclid.in's vectors (BASELINE configs[0]: LOOKUP_BITS 12):
    satisfied; the same witness with one cell altered is not, and the first violated row is the gate that holds the cell"""
    exe = os.path.join(ROOT, "halo2_vectordb_amd", "host", "example_mock")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "halo2_vectordb_amd", "csrc"), "../host/example_mock"])
    a, b = [0.123, 0.456, 1.789, 1.123], [1.123, 0.456, 0.789, 0.123]
    txt = "4 " + " ".join(repr(x) for x in a + b) + "\n"
    qa, qb = O.quantize(np.array(a)), O.quantize(np.array(b))
    for metric, name in ((0, "euclidean"), (1, "cosine"), (2, "manhattan"), (3, "hamming")):
        out = subprocess.run([exe, str(metric), "12"], input=txt, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, (out.stdout, out.stderr)
        c = O.Ctx(store=True, keygen=True)
        r = c.distance(name, qa, qb, L=12)
        head = out.stdout.splitlines()[0].split()
        assert int(head[1]) == len(c) and int(head[3]) == c.n_lookup and abs(float(head[5]) - float(O.dequantize(r))) < 1e-9
        assert "gate_rows_violated 0 lookup_cells_out_of_table 0 lookup_copies_unequal 0 constants_changed 0" in out.stdout
    bad = subprocess.run([exe, "0", "12", "1003"], input=txt, capture_output=True, text=True, timeout=120)
    assert bad.returncode == 4, (bad.stdout, bad.stderr)
    rep = bad.stdout.splitlines()[1].split()
    assert int(rep[1]) >= 1 and 1000 <= int(rep[9]) <= 1003


def test_example_fixed_point_in_compiled_code(O):
    """host/example_fixed_point.cpp = examples/fixed_point.rs:38-112 (FixedPointChip<32>: load_witness(x), qexp2, qlog2 when x > 0, qsin,
    all public) through the C++ mirror of FixedPointInstructions: the context's cell counts are the oracle's for the same calls, the
    values the oracle's to the printed precision, and the errors against f64 are what fixed-point at 32 fractional bits allows — at the
    inputs the example's own comments list (examples/fixed_point.rs:121-128)"""
    import math
    exe = os.path.join(ROOT, "halo2_vectordb_amd", "host", "example_fixed_point")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "halo2_vectordb_amd", "csrc"), "../host/example_fixed_point"])
    P, L = 32, 12
    for x in (-12.0, -1.88724767676867, 0.0, 1.0, 1.128, 2.0, 4.0, 0.25 * math.pi):
        out = subprocess.run([exe, str(L), repr(x)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, (x, out.stdout, out.stderr)
        lines = out.stdout.splitlines()
        c = O.Ctx(store=True, keygen=True)
        q = O.quantize(np.array([x]), P)
        c.assign_witnesses(q)
        want = [("exp2", c.op("qexp2", q[0], P=P, L=L), 2.0 ** x)]
        if x > 0:
            want.append(("log2", c.op("qlog2", q[0], P=P, L=L), math.log2(x)))
        want.append(("sin", c.op("qsin", q[0], P=P, L=L), math.sin(x)))
        head = lines[0].split()
        assert (int(head[1]), int(head[3]), int(head[5])) == (len(c), c.n_lookup, 1 + len(want)), (x, lines[0])
        assert len(lines) == 1 + len(want)
        for line, (name, value, native) in zip(lines[1:], want):
            got = line.split()
            assert got[0] == name
            zk = float(O.dequantize(value.reshape(1, 4), P)[0])
            if abs(zk) < 1e12:                                # (sin(0) is -1 ulp: dequantization's quirk, printed as it is)
                assert abs(float(got[1]) - zk) <= 1e-9 * max(1.0, abs(zk)), (x, line)
                # (the example prints the error and asserts nothing; at its 32 fractional bits the sine's 14 Horner steps keep ~1e-5 and the
                #  logarithm's alternating coefficients ~2e-3: the algorithm's accuracy, the same in the oracle)
                assert abs(zk - native) <= 5e-3 * max(1.0, abs(native)), (x, name, zk, native)
            assert abs(float(got[2]) - native) <= 1e-9 * max(1.0, abs(native))
