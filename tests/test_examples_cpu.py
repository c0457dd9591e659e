"""The reference's example workloads on its own input files, on the CPU restatement (oracle): every example with the
parameters it hard-codes (examples/*.rs) and the LOOKUP_BITS / degree its README gives, checked the way the reference's own
tests check a chip (dequantized result against an f64 recomputation at relative 1e-6, exact cluster ids / index) and the way
its `mock` stage would (every gate row a + b c = d and every lookup cell in range).  The reference records no expected
outputs for these inputs, so layout-level parity stays unpinned; what these tests pin is the value level, on the
reference-held inputs.  The GPU counterpart (GPU == oracle bit for bit on the same inputs) is tests/test_gpu_examples.py."""
import numpy as np

import examples_common as E


def test_distances_in(O):
    r = E.oracle_distances(O)
    d = E.load("distances")
    for (m, _), v in r["results"].items():
        got, want = float(O.dequantize(v)), E.F64[m](d["a"], d["b"])
        assert abs(got - want) <= (1e-6 if m != "cosine" else 1e-7) * max(abs(want), 1.0), (m, got, want)
    # hand-derived exact values: Manhattan distance is exactly 2 * 2^48
    assert O.fr_to_ints(r["results"][("manhattan", 1)])[0] == 2 << 48
    c = r["ctx"]
    assert c.check_gates(12) == 0
    # README: -k 13 with LOOKUP_BITS = 12 must be enough rows for the four distances
    assert (len(c.break_points()) + 1) >= 1 and c.n_lookup < (len(c.break_points()) + 1) * (8192 - 9)


def test_euclid_in(O):
    r = E.oracle_distances(O, "euclid")
    d = E.load("euclid")
    vals = [float(O.dequantize(v)) for v in r["results"].values()]
    assert len(vals) == 10 and len(set(vals)) == 1 and E.rel_close(vals[0], E.f64_euclidean(d["a"], d["b"]))
    assert r["ctx"].check_gates(12) == 0   # distance > 0: the qsqrt(0) defect examples/euclid.rs:25 hunts does not occur here


def _check_kmeans(O, name):
    r = E.oracle_kmeans(O, name)
    c = r["ctx"]
    assert c.err == 0
    d = E.load(name)
    cent, ids = E.f64_kmeans(d["vectors"], 4, 10, E.f64_cosine)
    got = O.dequantize(r["centroids"].reshape(-1, 4)).reshape(4, -1)
    assert E.rel_close(got, cent)
    ind = O.dequantize(r["indicators"].reshape(-1, 4)).reshape(len(ids), 4)
    assert [int(np.argmax(x == 1.0)) for x in ind] == ids           # tests/vectordb/mod.rs:121-132
    # the cosine variant is the satisfiable one (examples/kmeans.rs:48-49): every gate row and lookup cell holds
    assert c.check_gates(E.README["kmeans"]["L"]) == 0
    return r


def test_kmeans_in(O):
    r = _check_kmeans(O, "kmeans")
    # 20 vectors = 5 copies of 4 distinct rows: every cluster is one row and its copies, the centroids are those rows
    d = np.array(E.load("kmeans")["vectors"])
    assert E.rel_close(O.dequantize(r["centroids"].reshape(-1, 4)).reshape(4, -1), d[:4])
    assert len(r["ctx"].break_points()) + 1 == 546      # advice columns at k = 16 (README.md:78-79)


def test_kmeans_big_in(O):
    _check_kmeans(O, "kmeans_big")


def test_merkle_in(O):
    r = E.oracle_merkle(O)
    assert np.array_equal(r["root"], O.poseidon_merkle_root(r["qv"]))
    assert r["ctx"].check_gates(12) == 0
    # 3 leaves of 3 words (one full chunk + one chunk of a single word, which carries the padding itself: 2256 + 2253 cells)
    # padded to 4 leaves with the shared zero cell, then 3 tree nodes of a full chunk + the padding-only permutation
    assert len(r["ctx"]) - r["offset"] == 3 * (2256 + 2253) + 1 + 3 * (2256 + 2250)


def test_query_in(O):
    r = E.oracle_query(O)
    d = E.load("query")
    idx, want = E.f64_nearest(d["query"], d["database"], E.f64_cosine)
    ind = O.fr_to_ints(r["indicator"])
    # the database holds every row five times: all five copies of the nearest row tie, raw 0/1 indicators
    # (src/gadget/vectordb.rs:146-149), the first of them is the f64 answer (tests/vectordb/mod.rs:240-243)
    assert ind.index(1) == idx and sum(ind) == 5 and set(ind) == {0, 1}
    assert all(d["database"][i] == d["database"][idx] for i, v in enumerate(ind) if v)
    assert E.rel_close(O.dequantize(r["result"]), want)
    assert np.array_equal(r["root"], O.poseidon_merkle_root(r["qdb"]))
    # with ties select_by_indicator's witness ("the value at the last set indicator") no longer satisfies its own
    # accumulation gate acc + v * ind = acc' [UPSTREAM-RECALL halo2-base flex_gate]: 4 extra ties x 3 dimensions
    assert r["ctx"].check_gates(12) == 12


def test_poseidon_in(O):
    r = E.oracle_poseidon(O)
    assert np.array_equal(r["hash"], O.poseidon_hash_many(r["xy"].reshape(1, 2, 4))[0])
    assert r["ctx"].check_gates(12) == 0 and len(r["ctx"]) == 2 + 3 + 2256 + 2250
