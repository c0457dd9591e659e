"""The symbolic copy-constraint map of the fixed-point circuits (halo2_vectordb_amd/circuit_sym.py) against real witnesses
of the CPU restatement: every gate flag, every copy (an `Existing` cell equals the cell it copies), every constant, every
lookup source and every asserted constant must agree with the values of streams the oracle produced independently — on two
different inputs each, for all three metrics, nearest_vector and k-means — and the numpy block builders used at BASELINE
sizes must give exactly the whole-circuit trace.  (GPU witnesses go through the same checks in tests/test_gpu_copymap.py.)"""
import numpy as np
import pytest

from halo2_vectordb_amd import circuit_sym as CS


def to_ints(O, a):
    c = O.fr_to_canonical(np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4))
    return c[:, 0].astype(object) + (c[:, 1].astype(object) << 64) + (c[:, 2].astype(object) << 128) + (c[:, 3].astype(object) << 192)


def clean(rep, allow_asserts=False):
    bad = {k: v for k, v in rep.items() if v and not (allow_asserts and k == "asserts_violated")}
    return not bad, bad


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "manhattan", "hamming"])
@pytest.mark.parametrize("dim,P,L", [(3, 48, 12), (5, 32, 10), (1, 48, 13)])
def test_distance_map_on_oracle_witnesses(O, metric, dim, P, L):
    cm, outs = CS.trace_distance(metric, dim, P, L)
    rng = np.random.default_rng(dim * 7 + L)
    for trial in range(2):
        a, b = rng.uniform(-3, 3, dim), rng.uniform(-3, 3, dim)
        if trial and metric == "hamming":
            b[0] = a[0]                                       # one equal pair: what the Hamming distance counts
        qa, qb = O.quantize(a, P), O.quantize(b, P)
        c = O.Ctx(store=True, keygen=True)
        c.assign_witnesses(qa)
        c.assign_witnesses(qb)
        res = c.distance(metric, qa, qb, P=P, L=L)
        assert len(c) == cm.n_cells and c.n_lookup == len(cm.lookup_src)
        vals = to_ints(O, c.advice())
        ok, bad = clean(cm.check_witness(vals, to_ints(O, c.lookup()), flags=c.selectors()))
        assert ok, bad
        assert vals[outs[0]] == to_ints(O, res)[0]
    # most cells are tied to something: copies, constants, asserted constants
    tied = (cm.copy_of != np.arange(cm.n_cells)) | (cm.const_idx >= 0)
    assert tied.mean() > 0.4 and cm.asserted.sum() == (0 if metric in ("manhattan", "hamming") else (8 if metric == "euclidean" else 16))   # per qlog2: is_invalid, 3 x 2 in check_power_of_two, the bracket


def test_distances_example_map_on_the_oracles_witness(O):
    """examples/distances.rs:40-59: the four distances of the same two assigned vectors in one context; the map's outputs are the cells
    the example makes public"""
    metrics = ("euclidean", "manhattan", "cosine", "hamming")
    cm, outs = CS.trace_distances(metrics, 3, 48, 12)
    qa, qb = O.quantize(np.array([0.123, 0.456, 1.789])), O.quantize(np.array([1.123, 0.456, 0.789]))
    c = O.Ctx(store=True, keygen=True)
    c.assign_witnesses(qa)
    c.assign_witnesses(qb)
    res = [c.distance(m, qa, qb, P=48, L=12) for m in metrics]
    assert len(c) == cm.n_cells and c.n_lookup == len(cm.lookup_src)
    vals = to_ints(O, c.advice())
    ok, bad = clean(cm.check_witness(vals, to_ints(O, c.lookup()), flags=c.selectors()))
    assert ok, bad
    assert [vals[o] for o in outs] == [to_ints(O, r)[0] for r in res]
    # every distance reads the assigned vectors themselves: each of the six input cells is copied by all four
    assert all(int((cm.copy_of == cell).sum()) >= 5 for cell in range(6))


def test_sqrt_of_zero_violates_its_asserted_constant(O):
    """the defect examples/euclid.rs:25 hunts: euclidean distance 0 -> qlog2(0): `is_invalid` is one but assert_is_const ties it to
    zero (fixed_point.rs:742-745), and the bracket 2^n <= a < 2^(n+1) cannot hold either.  Gates, copies and lookups are all
    satisfied — only the asserted constants expose it."""
    cm, _ = CS.trace_distance("euclidean", 3, 48, 12)
    q = O.quantize(np.array([0.5, 1.25, 2.0]), 48)
    c = O.Ctx(store=True, keygen=True)
    c.assign_witnesses(q)
    c.assign_witnesses(q)
    c.distance("euclidean", q, q, P=48, L=12)
    rep = cm.check_witness(to_ints(O, c.advice()), to_ints(O, c.lookup()), flags=c.selectors())
    ok, bad = clean(rep, allow_asserts=True)
    assert ok, bad
    assert rep["asserts_violated"] >= 1 and c.check_gates(12) == 0


@pytest.mark.parametrize("metric,n,dim", [("euclidean", 3, 2), ("cosine", 2, 3), ("manhattan", 4, 2), ("euclidean", 1, 2)])
def test_nearest_map_and_block_builder(O, metric, n, dim):
    P, L = 48, 11
    cm, (ind, res) = CS.trace_nearest(metric, n, dim, P, L)
    rng = np.random.default_rng(n * 31 + dim)
    q, db = O.quantize(rng.random(dim) + 0.1), O.quantize(rng.random((n, dim)) + 0.1)
    c = O.Ctx(store=True, keygen=True)
    c.assign_witnesses(q)
    c.assign_witnesses(db)
    c.nearest_vector(metric, q, db, P=P, L=L)
    assert len(c) == cm.n_cells
    ok, bad = clean(cm.check_witness(to_ints(O, c.advice()), to_ints(O, c.lookup()), flags=c.selectors()))
    assert ok, bad
    bm, (bind, bres) = CS.build_nearest(metric, n, dim, P, L)
    for name in ("copy_of", "const_idx", "asserted", "gate", "lookup_src"):
        a, b = getattr(cm, name), getattr(bm, name)
        if name == "const_idx":      # constants are numbered in order of first use by either construction: compare the values
            a = np.where(a >= 0, np.asarray(cm.consts + [0], dtype=object)[a], -1)
            b = np.where(b >= 0, np.asarray(bm.consts + [0], dtype=object)[b], -1)
        assert np.array_equal(a, b), name
    assert list(bind) == ind and list(bres) == res


@pytest.mark.parametrize("metric,n,dim,K,I", [("euclidean", 4, 2, 2, 2), ("cosine", 5, 3, 2, 1), ("manhattan", 6, 2, 3, 2)])
def test_kmeans_map_and_block_builder(O, metric, n, dim, K, I):
    P, L = 48, 11
    cm, (cent, inds) = CS.trace_kmeans(metric, n, dim, K, I, P, L)
    rng = np.random.default_rng(n + dim + K)
    for trial in range(2):
        while True:
            v = O.quantize(rng.random((n, dim)) * (1 + 3 * trial) + 0.05)
            c = O.Ctx(store=True, keygen=True)
            c.assign_witnesses(v)
            c.kmeans(metric, v, K, I, P=P, L=L)
            if c.err == 0:
                break
        assert len(c) == cm.n_cells and c.n_lookup == len(cm.lookup_src)
        rep = cm.check_witness(to_ints(O, c.advice()), to_ints(O, c.lookup()), flags=c.selectors())
        # Euclidean k-means is unsatisfiable at iteration 0 (a vector is its own centroid: qsqrt(0)); nothing else may be off
        ok, bad = clean(rep, allow_asserts=(metric == "euclidean"))
        assert ok, bad
        assert (rep["asserts_violated"] > 0) == (metric == "euclidean")
    bm, (bcent, bind) = CS.build_kmeans(metric, n, dim, K, I, P, L)
    for name in ("copy_of", "const_idx", "asserted", "gate", "lookup_src"):
        a, b = getattr(cm, name), getattr(bm, name)
        if name == "const_idx":
            a = np.where(a >= 0, np.asarray(cm.consts + [0], dtype=object)[a], -1)
            b = np.where(b >= 0, np.asarray(bm.consts + [0], dtype=object)[b], -1)
        assert np.array_equal(a, b), name
    assert np.array_equal(bcent, np.asarray(cent)) and np.array_equal(bind, np.asarray(inds))


def test_an_altered_witness_is_caught_by_the_map(O):
    """what the gates alone miss: the output cell of a qmul replaced consistently inside its own gate (the product cell and the
    division cells that follow are recomputed by a cheating prover) still differs from the copies later gates hold"""
    cm, _ = CS.trace_distance("euclidean", 2, 48, 12)
    q1, q2 = O.quantize(np.array([0.5, 1.5])), O.quantize(np.array([1.25, 0.25]))
    c = O.Ctx(store=True, keygen=True)
    c.assign_witnesses(q1)
    c.assign_witnesses(q2)
    c.distance("euclidean", q1, q2, P=48, L=12)
    vals = to_ints(O, c.advice())
    lk = to_ints(O, c.lookup())
    assert clean(cm.check_witness(vals, lk))[0]
    # the first qsub output (cell 4, a fresh witness) is copied into the inner product: change the copy only
    users = np.flatnonzero(cm.copy_of == 4)
    assert users.size >= 2
    bad = vals.copy()
    bad[users[0]] = (bad[users[0]] + 1) % CS.R
    assert cm.check_witness(bad, lk)["copies_unequal"] >= 1


@pytest.mark.parametrize("x", [1.128, -1.88724767676867, 0.7853981633974483, 4.0])
def test_fixed_point_example_map_on_the_oracles_witness(O, x):
    """examples/fixed_point.rs:55-111 at its PRECISION_BITS = 32: load_witness(x), qexp2, qlog2 for a positive x, qsin; x and every result
    public — the symbolic map (qmod with its asserted sign, the sine's two polynomials) on the oracle's cells"""
    P, L = 32, 12
    ops = ("qexp2",) + (("qlog2",) if x > 0 else ()) + ("qsin",)
    cm, outs = CS.trace_fixed_point(ops, P, L)
    q = O.quantize(np.array([x]), P)
    c = O.Ctx(store=True, keygen=True)
    c.assign_witnesses(q)
    res = [c.op(name, q[0], P=P, L=L) for name in ops]
    assert len(c) == cm.n_cells and c.n_lookup == len(cm.lookup_src) and c.err == 0
    vals = to_ints(O, c.advice())
    ok, bad = clean(cm.check_witness(vals, to_ints(O, c.lookup()), flags=c.selectors()))
    assert ok, bad
    assert [vals[o] for o in outs] == [to_ints(O, q)[0]] + [to_ints(O, r)[0] for r in res]


@pytest.mark.parametrize("name", ["neg", "qabs", "is_neg", "qsqrt", "qlog2", "qexp2", "qlog", "qexp", "sign", "clip", "qsin", "qcos", "qtan", "qsinh", "qcosh", "qtanh"])
def test_every_unary_fixed_point_operation_map_on_the_oracles_witness(O, name):
    """the symbolic map of one FixedPointInstructions call on a loaded witness (what FixedPointHotPath(ops=...) proves), on the oracle's
    cells, for a positive and — where the operation takes one — a negative operand"""
    P, L = 48, 12
    cm, outs = CS.trace_fixed_point((name,), P, L)
    for x in ((1.375, 0.6) if name in ("qsqrt", "qlog2", "qlog") else (1.375, -2.25)):
        q = O.quantize(np.array([x]), P)
        c = O.Ctx(store=True, keygen=True)
        c.assign_witnesses(q)
        res = c.op(name, q[0], P=P, L=L)
        assert len(c) == cm.n_cells and c.n_lookup == len(cm.lookup_src) and c.err == 0
        vals = to_ints(O, c.advice())
        ok, bad = clean(cm.check_witness(vals, to_ints(O, c.lookup()), flags=c.selectors()))
        assert ok, (name, x, bad)
        assert vals[outs[1]] == to_ints(O, res)[0]
