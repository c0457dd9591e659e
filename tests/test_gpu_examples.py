"""The reference's example workloads on its own input files (tests/golden/reference_data = /root/reference/data/*.in), on
the GPU through the C ABI: the cells the HIP kernels emit equal the CPU restatement's bit for bit (advice stream, lookup
stream, gate-start bits), and the dequantized results meet the reference tests' tolerance against f64.  Parameters are the
examples' own (examples/*.rs) with the README's LOOKUP_BITS / degree.  CPU-only counterpart: tests/test_examples_cpu.py."""
import numpy as np
import pytest

import examples_common as E

pytestmark = pytest.mark.gpu
N_BLIND = 7


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init()
    return a


def same_cells(got, c, off, n_adv=None, n_lk=None):
    """the GPU's gadget cells against the slice of the oracle's context that starts at `off` = (advice, lookup) offsets"""
    adv, lk, sel = c.advice(), c.lookup(), c.selectors()
    oa, ol = off if isinstance(off, tuple) else (off, 0)
    n_adv = got["stream"].shape[0] if n_adv is None else n_adv
    bad = np.nonzero((got["stream"] != adv[oa:oa + n_adv]).any(axis=1))[0]
    assert bad.size == 0, f"first differing advice cells {bad[:5]}"
    if got.get("flags") is not None:
        assert np.array_equal(got["selectors"], sel[oa:oa + n_adv] & 1)
    if "lookup" in got:
        n_lk = got["lookup"].shape[0] if n_lk is None else n_lk
        assert np.array_equal(got["lookup"], lk[ol:ol + n_lk])


def test_distances_in(api, O):
    """examples/distances.rs on data/distances.in: euclidean, manhattan, cosine, hamming"""
    r = E.oracle_distances(O)
    d, L = E.load("distances"), E.README["distances"]["L"]
    qa, qb = api.quantize([d["a"]]), api.quantize([d["b"]])
    assert np.array_equal(qa[0], r["qa"]) and np.array_equal(qb[0], r["qb"])
    for (m, i), want in r["results"].items():
        got = api.wit_distance(m, qa, qb, L=L, selectors=True)
        assert np.array_equal(got["result"][0], want)
        same_cells(got, r["ctx"], r["offsets"][(m, i)])
        f = E.F64[m](d["a"], d["b"])
        assert abs(float(api.dequantize(got["result"])[0]) - f) <= 1e-6 * max(abs(f), 1.0)


def test_euclid_in(api, O):
    """examples/euclid.rs: ten euclidean distances of the same pair = ten identical instances one after the other"""
    r = E.oracle_distances(O, "euclid")
    d = E.load("euclid")
    qa, qb = api.quantize([d["a"]] * 10), api.quantize([d["b"]] * 10)
    got = api.wit_distance("euclidean", qa, qb, L=12, selectors=True)
    same_cells(got, r["ctx"], (6, 0))
    assert got["stream"].shape[0] == len(r["ctx"]) - 6
    assert E.rel_close(api.dequantize(got["result"]), [E.f64_euclidean(d["a"], d["b"])] * 10)


@pytest.mark.parametrize("name", ["kmeans", "kmeans_big"])
def test_kmeans_in(api, O, name):
    """examples/kmeans.rs: kmeans::<4, 10> with the cosine distance, PRECISION_BITS 48, LOOKUP_BITS 15, k = 16: the cells, the
    column layout and sampled commitments of the resident hot path against the oracle, results against f64"""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    r = E.oracle_kmeans(O, name)
    c, cfg = r["ctx"], E.README["kmeans"]
    vec = np.array(E.load(name)["vectors"], dtype=np.float64)
    got = api.wit_kmeans("cosine", api.quantize(vec), 4, 10, L=cfg["L"], selectors=True)
    assert np.array_equal(got["centroids"], r["centroids"]) and np.array_equal(got["indicators"], r["indicators"])
    same_cells(got, c, r["offset"])
    assert got["stream"].shape[0] == len(c) - r["offset"][0] and got["lookup"].shape[0] == c.n_lookup
    del got
    cent, ids = E.f64_kmeans(vec, 4, 10, E.f64_cosine)
    assert E.rel_close(api.dequantize(r["centroids"].reshape(-1, 4)).reshape(4, -1), cent)
    # the Prove arm's hot path on the same input file
    hp = KmeansHotPath(n=vec.shape[0], dim=vec.shape[1], K=4, I=10, k=cfg["k"], P=48, L=cfg["L"], metric="cosine", vectors=vec, tau=0xE4).setup()
    assert hp.n_cells == len(c) and hp.n_lookup == c.n_lookup and np.array_equal(hp.bp, c.break_points())
    com = hp.step().copy()
    gc, gi = hp.results()
    assert np.array_equal(gc, r["centroids"]) and np.array_equal(gi, r["indicators"])
    hp.relayout()
    pick = [0, 1, hp.n_adv_cols // 2, hp.n_adv_cols - 1, hp.n_adv_cols, hp.n_cols - 1]
    cols = hp.download_columns(pick)
    want_adv = O.layout_columns(c.advice(), hp.bp, cfg["k"], hp.n_adv_cols)
    want_lk = O.layout_lookup(c.lookup(), cfg["k"], hp.n_lk_cols)
    used = hp.rows - N_BLIND
    for j, col in enumerate(pick):
        w = want_adv[col] if col < hp.n_adv_cols else want_lk[col - hp.n_adv_cols]
        assert np.array_equal(cols[j][:used], w[:used]), f"column {col}"
    assert np.array_equal(com[pick], O.msm_batch(cols, hp.g_lagrange, threads=8))
    hp.free()


def test_merkle_in(api, O):
    """examples/merkle.rs: PRECISION_BITS 32; the trace, its root, the hash-only kernel's root"""
    r = E.oracle_merkle(O)
    qv = api.quantize(np.array(E.load("merkle")["vectors"]), 32)
    assert np.array_equal(qv, r["qv"])
    got = api.wit_merkle(qv, selectors=True)
    assert np.array_equal(got["root"], r["root"]) and np.array_equal(api.poseidon_merkle_root(qv), r["root"])
    same_cells(got, r["ctx"], r["offset"])
    assert got["stream"].shape[0] == len(r["ctx"]) - r["offset"]


def test_query_in(api, O):
    """examples/query.rs: nearest_vector with the cosine distance, then the database's Merkle commitment.  The database holds
    every row five times, so five indicators are set and the result is the tied rows' common value."""
    r = E.oracle_query(O)
    d, L = E.load("query"), E.README["query"]["L"]
    qq, qdb = api.quantize(np.array(d["query"])), api.quantize(np.array(d["database"]))
    nv = api.wit_nearest("cosine", qq, qdb, L=L, selectors=True)
    assert np.array_equal(nv["indicator"], r["indicator"]) and np.array_equal(nv["result"], r["result"])
    same_cells(nv, r["ctx"], r["off_nv"])
    assert r["off_nv"][0] + nv["stream"].shape[0] == r["off_mk"]
    mk = api.wit_merkle(qdb, selectors=True)
    assert np.array_equal(mk["root"], r["root"])
    same_cells(mk, r["ctx"], r["off_mk"])
    idx, want = E.f64_nearest(d["query"], d["database"], E.f64_cosine)
    ind = O.fr_to_ints(nv["indicator"])
    assert ind.index(1) == idx and sum(ind) == 5
    assert E.rel_close(api.dequantize(nv["result"]), want)


def test_poseidon_in(api, O):
    """examples/poseidon.rs on data/poseidon.in = ["6", "100"]"""
    r = E.oracle_poseidon(O)
    msg = r["xy"].reshape(1, 2, 4)
    assert np.array_equal(api.poseidon_hash_many(msg)[0], r["hash"])
    got = api.wit_merkle(msg, selectors=True)       # one leaf, no tree level: update([x, y]); squeeze
    assert np.array_equal(got["root"], r["hash"])
    same_cells(got, r["ctx"], r["offset"])
