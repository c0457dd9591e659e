"""GPU parity of the witness-stream generators (C ABI vdb_wit_*, vdb_layout_*) against the CPU oracle:
advice stream, lookup stream, gate-start bits and results must be bit-identical on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init()
    return a


def sift_like(rng, n, dim):
    v = rng.integers(0, 219, size=(n, dim)).astype(np.float64)
    v[v.sum(axis=1) == 0, 0] = 1.0
    return v


def oracle_distance(O, metric, qa, qb, L):
    c = O.Ctx(store=True, keygen=True)
    res = [c.distance(metric, a, b, L=L) for a, b in zip(qa, qb)]
    return c, np.stack(res)


def assert_streams(api_out, c):
    adv, lk, sel = c.advice(), c.lookup(), c.selectors()
    assert api_out["stream"].shape == adv.shape
    bad = np.nonzero((api_out["stream"] != adv).any(axis=1))[0]
    assert bad.size == 0, f"first differing advice cell {bad[:5]} of {adv.shape[0]}"
    if "lookup" in api_out:
        assert api_out["lookup"].shape == lk.shape
        badl = np.nonzero((api_out["lookup"] != lk).any(axis=1))[0]
        assert badl.size == 0, f"first differing lookup cell {badl[:5]}"
    if api_out.get("selectors") is not None:
        assert np.array_equal(api_out["selectors"], sel)


def test_quantize_dequantize_host(api, O):
    rng = np.random.default_rng(500)
    x = np.concatenate([rng.uniform(-300, 300, 200), [0.0, -0.0, 0.5, -0.5, 2.0 ** -49, 1e30, -1e30, np.inf, np.nan, 218.0]])
    for P in (32, 48, 63):
        q = api.quantize(x, P)
        assert np.array_equal(q, O.quantize(x, P))
        assert np.array_equal(api.dequantize(q, P), O.dequantize(q, P), equal_nan=True)


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "manhattan", "hamming"])
@pytest.mark.parametrize("L,dim", [(12, 4), (13, 10), (15, 128), (13, 1), (13, 130)])
def test_distance_streams(api, O, metric, L, dim):
    rng = np.random.default_rng(600 + L + dim)
    n = 5 if dim < 100 else 3
    a = rng.uniform(-3, 3, size=(n, dim))
    b = rng.uniform(-3, 3, size=(n, dim))
    if dim == 128:
        a, b = sift_like(rng, n, dim), sift_like(rng, n, dim)
    a[0] = b[0]  # distance 0: qsqrt(0) quirk path
    a[1, ::3] = b[1, ::3]  # some equal elements (what the Hamming distance counts)
    qa, qb = O.quantize(a), O.quantize(b)
    got = api.wit_distance(metric, qa, qb, L=L, selectors=True)
    c, res = oracle_distance(O, metric, qa, qb, L)
    assert np.array_equal(got["result"], res)
    assert_streams(got, c)


def test_distance_reference_data(api, O):
    # /root/reference/data/distances.in values (committed here as literals)
    a, b = [[0.123, 0.456, 1.789]], [[1.123, 0.456, 0.789]]
    qa, qb = api.quantize(a), api.quantize(b)
    got = api.wit_distance("euclidean", qa, qb, L=12)
    assert abs(float(api.dequantize(got["result"])[0]) - 2 ** 0.5) < 1e-6 * 2 ** 0.5
    man = api.wit_distance("manhattan", qa, qb, L=12)
    assert O.fr_to_ints(man["result"])[0] == 2 << 48
    ham = api.wit_distance("hamming", qa, qb, L=12)                       # one of three elements equal: 1 - 1/3
    assert abs(float(api.dequantize(ham["result"])[0]) - 2 / 3) < 1e-6


def test_distance_other_precision(api, O):
    rng = np.random.default_rng(601)
    a, b = rng.uniform(-2, 2, size=(2, 6)), rng.uniform(-2, 2, size=(2, 6))
    qa, qb = O.quantize(a, 32), O.quantize(b, 32)
    got = api.wit_distance("euclidean", qa, qb, P=32, L=10, selectors=True)
    c = O.Ctx(store=True, keygen=True)
    res = np.stack([c.distance("euclidean", x, y, P=32, L=10) for x, y in zip(qa, qb)])
    assert np.array_equal(got["result"], res)
    assert_streams(got, c)


@pytest.mark.parametrize("metric,n,dim", [("euclidean", 4, 4), ("euclidean", 9, 16), ("cosine", 5, 8), ("manhattan", 70, 3), ("hamming", 6, 5)])
def test_nearest_vector(api, O, metric, n, dim):
    rng = np.random.default_rng(700 + n)
    db = rng.random((n, dim))
    q = rng.random(dim)
    if n >= 9:
        db[5] = db[2]  # tie: both indicators set, select_by_indicator keeps the last
    qq, qdb = O.quantize(q), O.quantize(db)
    got = api.wit_nearest(metric, qq, qdb, selectors=True)
    c = O.Ctx(store=True, keygen=True)
    ind, res = c.nearest_vector(metric, qq, qdb)
    assert np.array_equal(got["indicator"], ind) and np.array_equal(got["result"], res)
    assert_streams(got, c)


@pytest.mark.parametrize("metric,n,dim,K,I", [("euclidean", 6, 3, 2, 2), ("cosine", 7, 5, 3, 2), ("manhattan", 30, 5, 2, 4), ("euclidean", 20, 128, 4, 2)])
def test_kmeans(api, O, metric, n, dim, K, I):
    rng = np.random.default_rng(800 + n)
    v = rng.random((n, dim)) + 0.05 if dim < 100 else sift_like(rng, n, dim)
    qv = O.quantize(v)
    L = 13 if dim < 100 else 15
    got = api.wit_kmeans(metric, qv, K, I, L=L, selectors=True)
    c = O.Ctx(store=True, keygen=True)
    cent, ind = c.kmeans(metric, qv, K, I, L=L)
    assert c.err == 0
    assert np.array_equal(got["centroids"], cent) and np.array_equal(got["indicators"], ind)
    assert_streams(got, c)


def test_kmeans_errors(api, O):
    qv = O.quantize(np.ones((3, 2)))
    with pytest.raises(api.VdbError) as e:
        api.wit_kmeans("euclidean", qv, 3, 1)  # K < N violated (vectordb.rs:238)
    assert e.value.code == -5


@pytest.mark.parametrize("n,dim", [(1, 3), (2, 4), (3, 5), (5, 2), (8, 128), (33, 7)])
def test_merkle_trace(api, O, n, dim):
    rng = np.random.default_rng(900 + n)
    v = O.quantize(rng.random((n, dim)), 32)
    got = api.wit_merkle(v, selectors=True)
    c = O.Ctx(store=True, keygen=True)
    root = c.merkle_commitment(v)
    assert np.array_equal(got["root"], root)
    assert np.array_equal(got["root"], api.poseidon_merkle_root(v))
    assert_streams(got, c)


def test_layout_plan_and_columns(api, O):
    rng = np.random.default_rng(1000)
    a, b = O.quantize(rng.random((3, 16))), O.quantize(rng.random((3, 16)))
    k = 10
    c = O.Ctx(store=True, keygen=True, plan_k=k)
    for x, y in zip(a, b):
        c.distance("euclidean", x, y)
    got = api.wit_distance("euclidean", a, b, selectors=True)
    bp = api.layout_plan(got["flags"], k)
    assert np.array_equal(bp, c.break_points())
    cols, lcols = api.layout_columns(got["stream"], bp, k, lookup=got["lookup"])
    want = O.layout_columns(c.advice(), c.break_points(), k, len(bp) + 1)
    assert np.array_equal(cols, want)
    wantl = O.layout_lookup(c.lookup(), k, lcols.shape[0])
    assert np.array_equal(lcols, wantl)


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "manhattan"])
def test_smallest_and_ragged_shapes(api, O, metric):
    """degenerate shapes through every gadget: one dimension, one database vector, one cluster, a single iteration,
    dimensions just past a multiple of the head kernel's block (257), an odd database size"""
    rng = np.random.default_rng(4242)
    # nearest: one vector, one dimension
    for n, dim in ((1, 1), (1, 5), (2, 1), (3, 257)):
        db, q = rng.random((n, dim)) + 0.1, rng.random(dim) + 0.1
        qq, qdb = O.quantize(q), O.quantize(db)
        got = api.wit_nearest(metric, qq, qdb, selectors=True)
        c = O.Ctx(store=True, keygen=True)
        ind, res = c.nearest_vector(metric, qq, qdb)
        assert np.array_equal(got["indicator"], ind) and np.array_equal(got["result"], res), (n, dim)
        assert_streams(got, c)
    # kmeans: a single cluster, a single iteration, one dimension
    for n, dim, K, I in ((2, 1, 1, 1), (3, 2, 1, 2), (5, 1, 2, 1), (7, 129, 3, 1)):
        v = rng.random((n, dim)) + 0.05
        qv = O.quantize(v)
        c = O.Ctx(store=True, keygen=True)
        cent, ind = c.kmeans(metric, qv, K, I)
        if c.err:   # e.g. cosine in one dimension: every vector is at distance 0 of the first centroid, the others stay empty
            with pytest.raises(api.VdbError) as e:   # the reference panics on the division by the empty cluster's size
                api.wit_kmeans(metric, qv, K, I, selectors=True)
            assert e.value.code == -5
            continue
        got = api.wit_kmeans(metric, qv, K, I, selectors=True)
        assert np.array_equal(got["centroids"], cent) and np.array_equal(got["indicators"], ind), (n, dim, K, I)
        assert_streams(got, c)
