"""Oracle pinning, part 3: the fixed-point / distance / vectordb gadgets.

Pins available from the reference itself:
  * fixed inputs data/distances.in with hand-derivable results;
  * tolerance pins of tests/distances_test.rs / tests/vectordb_test.rs / tests/demo_test.rs:
    chip result vs f64 at relative 1e-6 (assert_float_relative_eq default), P=48, LOOKUP_BITS=13;
    k-means cluster ids and nearest index exact.
Cell ORDER follows halo2-base templates [UPSTREAM-RECALL] => parity unpinned; we check instead that
every emitted gate row satisfies a + b*c = d and every lookup cell is in range (MockProver-like).
"""
import math

import numpy as np
import pytest

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
A = [0.123, 0.456, 1.789]  # /root/reference/data/distances.in
B = [1.123, 0.456, 0.789]


def rel_eq(a, b, eps=1e-6):
    return abs(a - b) <= eps * max(abs(a), abs(b)) or abs(a - b) < 1e-12


def test_quantization_known_values(O, PY):
    qa, qb = O.fr_to_ints(O.quantize(A)), O.fr_to_ints(O.quantize(B))
    assert qa == [0x1F7CED916873, 0x74BC6A7EF9DB, 0x1C9FBE76C8B44]
    assert qb == [0x11F7CED916873, 0x74BC6A7EF9DB, 0xC9FBE76C8B44]
    assert [(x - y) % R for x, y in zip(qa, qb)] == [R - (1 << 48), 0, 1 << 48]
    fp = PY.FixedPoint(48)
    edge = [0.0, -0.0, 0.5, -0.5, 2.0 ** -49, 3 * 2.0 ** -49, -(2.0 ** -49), 1e30, -1e30, float("inf"), float("nan"), 218.0]
    assert O.fr_to_ints(O.quantize(edge)) == [fp.quantize(x) for x in edge]
    assert O.fr_to_ints(O.quantize([1.5, -1.5], P=32)) == [3 << 31, R - (3 << 31)]


def test_dequantization_quirk(O, PY):
    # fixed_point.rs:124: negatives come back as -(|v| - 2) / 2^P
    fp = PY.FixedPoint(48)
    vals = [0, 1, 1 << 48, 5 << 47, R - (1 << 48), R - 3, R - (1 << 96)]
    got = O.dequantize(O.fr_from_ints(vals))
    assert list(got) == [fp.dequantize(v) for v in vals]
    assert got[4] == -((1 << 48) - 2) / 2.0 ** 48


def test_distances_in_file(O):
    qa, qb = O.quantize(A), O.quantize(B)
    c = O.Ctx(store=True, keygen=True)
    d2 = c.inner_product(O.fr_sub(qa, qb), O.fr_sub(qa, qb))
    assert O.fr_to_ints(d2)[0] == 2 << 48  # dist_square = 2 * 2^48 exactly
    e = O.dequantize(c.distance("euclidean", qa, qb))
    assert rel_eq(float(e), math.sqrt(2.0))
    m = c.distance("manhattan", qa, qb)
    assert O.fr_to_ints(m)[0] == 2 << 48
    cs = O.dequantize(c.distance("cosine", qa, qb))
    na, nb = math.sqrt(sum(x * x for x in A)), math.sqrt(sum(x * x for x in B))
    assert rel_eq(float(cs), 1 - sum(x * y for x, y in zip(A, B)) / (na * nb))
    h = O.dequantize(c.distance("hamming", qa, qb))
    assert rel_eq(float(h), 1 - 1 / 3)
    assert c.err == 0 and c.check_gates(13) == 0


@pytest.mark.parametrize("L", [12, 13, 15])
def test_random_distances_like_reference_tests(O, PY, L):
    # tests/distances_test.rs: random_vector(10) in [0,1), chip vs f64 at rel 1e-6
    rng = np.random.default_rng(100 + L)
    fp = PY.FixedPoint(48)
    for _ in range(3):
        a, b = rng.random(10), rng.random(10)
        qa, qb = O.quantize(a), O.quantize(b)
        c = O.Ctx(store=True, keygen=True)
        e = c.distance("euclidean", qa, qb, L=L)
        assert rel_eq(float(O.dequantize(e)), float(np.linalg.norm(a - b)))
        assert O.fr_to_ints(e)[0] == fp.euclidean(O.fr_to_ints(qa), O.fr_to_ints(qb))
        cs = c.distance("cosine", qa, qb, L=L)
        assert rel_eq(float(O.dequantize(cs)), 1 - float(a @ b) / float(np.linalg.norm(a) * np.linalg.norm(b)))
        assert O.fr_to_ints(cs)[0] == fp.cosine(O.fr_to_ints(qa), O.fr_to_ints(qb))
        m = c.distance("manhattan", qa, qb, L=L)
        assert rel_eq(float(O.dequantize(m)), float(np.abs(a - b).sum()))
        assert O.fr_to_ints(m)[0] == fp.manhattan(O.fr_to_ints(qa), O.fr_to_ints(qb))
        b2 = b.copy()
        b2[::3] = a[::3]                                      # four of ten elements equal (random_vector never produces one)
        qb2 = O.quantize(b2)
        h = c.distance("hamming", qa, qb2, L=L)
        assert rel_eq(float(O.dequantize(h)), 0.6)
        assert O.fr_to_ints(h)[0] == fp.hamming(O.fr_to_ints(qa), O.fr_to_ints(qb2))
        assert O.fr_to_ints(c.distance("hamming", qa, qb, L=L))[0] == fp.hamming(O.fr_to_ints(qa), O.fr_to_ints(qb)) == fp.quantize(1.0)
        assert c.err == 0 and c.check_gates(L) == 0


def test_ops_vs_python_and_f64(O, PY):
    fp = PY.FixedPoint(48)
    rng = np.random.default_rng(11)
    c = O.Ctx(store=True, keygen=True)
    for _ in range(6):
        x, y = float(rng.uniform(-50, 50)), float(rng.uniform(-50, 50))
        qx, qy = O.quantize([x])[0], O.quantize([y])[0]
        ix, iy = O.fr_to_ints(qx)[0], O.fr_to_ints(qy)[0]
        assert O.fr_to_ints(c.op("qmul", qx, qy))[0] == fp.qmul(ix, iy)
        assert O.fr_to_ints(c.op("qdiv", qx, qy))[0] == fp.qdiv(ix, iy)
        assert O.fr_to_ints(c.op("qmin", qx, qy))[0] == fp.qmin(ix, iy)
        assert O.fr_to_ints(c.op("qabs", qx))[0] == fp.qabs(ix)
        assert O.fr_to_ints(c.op("is_neg", qx))[0] == fp.is_neg(ix)
        px = abs(x) + 0.01
        qp = O.quantize([px])[0]
        ip = O.fr_to_ints(qp)[0]
        assert O.fr_to_ints(c.op("qlog2", qp))[0] == fp.qlog2(ip)
        assert O.fr_to_ints(c.op("qsqrt", qp))[0] == fp.qsqrt(ip)
        assert rel_eq(float(O.dequantize(c.op("qsqrt", qp))), math.sqrt(px), 1e-6)
        assert rel_eq(float(O.dequantize(c.op("qlog2", qp))), math.log2(px), 1e-6) or abs(math.log2(px)) < 1e-3
        sx = x / 10
        qs = O.quantize([sx])[0]
        assert O.fr_to_ints(c.op("qexp2", qs))[0] == fp.qexp2(O.fr_to_ints(qs)[0])
        assert rel_eq(float(O.dequantize(c.op("qexp2", qs))), 2.0 ** sx, 1e-6)
    assert c.err == 0 and c.check_gates(13) == 0


def test_qmul_floors_qdiv_truncates(O):
    # SURVEY App. E item 4
    c = O.Ctx()
    a, b = O.fr_from_ints([R - 3])[0], O.fr_from_ints([1 << 47])[0]  # -3 * 0.5 in raw units
    assert O.fr_to_ints(c.op("qmul", a, b))[0] == R - 2  # floor(-1.5) = -2
    num, den = O.fr_from_ints([R - 3])[0], O.fr_from_ints([2 << 48])[0]
    assert O.fr_to_ints(c.op("qdiv", num, den))[0] == R - 1  # trunc(-1.5) = -1


def test_qsqrt_of_zero_quirk(O):
    # qsqrt(0) yields a tiny positive value; constraints are violated
    c = O.Ctx(store=True, keygen=True)
    z = O.fr_from_ints([0])[0]
    v = float(O.dequantize(c.op("qsqrt", z)))
    assert 0 < v < 1e-6


def test_cell_counts_match_survey_model(O):
    # SURVEY App. B rows that do not depend on idx_to_indicator
    want = {12: dict(is_neg=(160, 46), qabs=(172, 46), qmul=(294, 82), signed_div_scale=(290, 82)),
            13: dict(is_neg=(148, 42), qabs=(160, 42), qmul=(270, 74), signed_div_scale=(266, 74)),
            15: dict(is_neg=(130, 36), qabs=(142, 36), qmul=(240, 64), signed_div_scale=(236, 64))}
    x, y = O.quantize([3.25])[0], O.quantize([-1.5])[0]
    for L, ops in want.items():
        for name, (adv, lk) in ops.items():
            c = O.Ctx()
            c.op(name, x, y, L=L)
            assert (len(c), c.n_lookup) == (adv, lk), (L, name)


def test_cell_counts_data_independent(O):
    rng = np.random.default_rng(12)
    sizes = set()
    for _ in range(4):
        a, b = O.quantize(rng.uniform(-5, 5, 6)), O.quantize(rng.uniform(-5, 5, 6))
        c = O.Ctx(store=True, keygen=True)
        c.distance("euclidean", a, b)
        sizes.add((len(c), c.n_lookup, bytes(c.selectors())))
    assert len(sizes) == 1


def test_nearest_vector_like_reference_test(O, PY):
    # tests/vectordb_test.rs:31-42: 4 x 4-dim, index exact, vector rel-eq
    rng = np.random.default_rng(13)
    fp = PY.FixedPoint(48)
    q, db = rng.random(4), rng.random((4, 4))
    qq, qdb = O.quantize(q), O.quantize(db)
    c = O.Ctx(store=True, keygen=True)
    ind, res = c.nearest_vector("euclidean", qq, qdb)
    want = int(np.argmin(np.linalg.norm(db - q, axis=1)))
    assert [int(v) for v in O.fr_to_ints(ind)] == [1 if i == want else 0 for i in range(4)]
    assert np.allclose(O.dequantize(res), db[want], rtol=1e-6)
    pi, pr = fp.nearest_vector(O.fr_to_ints(qq), [O.fr_to_ints(v) for v in qdb], fp.euclidean)
    assert pi == [int(v) for v in O.fr_to_ints(ind)] and pr == O.fr_to_ints(res)
    assert c.err == 0 and c.check_gates(13) == 0


def test_kmeans_like_reference_test(O, PY):
    # tests/vectordb_test.rs:12-29 shape (kmeans<2,4> on 5-dim vectors), fewer vectors for speed
    rng = np.random.default_rng(14)
    fp = PY.FixedPoint(48)
    vecs = rng.random((8, 5))
    qv = O.quantize(vecs)
    c = O.Ctx(store=True, keygen=False)
    cent, ind = c.kmeans("euclidean", qv, 2, 4)
    assert c.err == 0
    pc, pind = fp.kmeans([O.fr_to_ints(v) for v in qv], 2, 4, fp.euclidean)
    assert [O.fr_to_ints(x) for x in cent] == pc
    assert [O.fr_to_ints(x) for x in ind] == pind
    # f64 k-means (tests/vectordb/mod.rs:31-91)
    cf = vecs[:2].copy()
    for _ in range(4):
        ids = np.array([int(np.argmin([np.linalg.norm(v - cc) for cc in cf])) for v in vecs])
        cf = np.array([vecs[ids == k].mean(axis=0) for k in range(2)])
    assert np.allclose(O.dequantize(cent), cf, rtol=1e-6)
    got_ids = [[float(x) for x in O.dequantize(i)].index(1.0) for i in ind]
    assert got_ids == list(ids)


def test_division_by_zero_is_error(O):
    # BigUint division by zero panics upstream (empty cluster in kmeans,
    # all-zero vector under cosine); the oracle flags err instead of aborting
    c = O.Ctx()
    c.op("qdiv", O.quantize([1.0])[0], O.quantize([0.0])[0])
    assert c.err != 0
    # an all-zero vector under cosine does NOT divide by zero: qsqrt(0) is a tiny positive number
    # (test_qsqrt_of_zero_quirk), so the denominator is non-zero and the witness continues
    c = O.Ctx()
    c.distance("cosine", O.quantize([0.0, 0.0]), O.quantize([1.0, 2.0]))
    assert c.err == 0


def test_break_points_and_layout(O):
    rng = np.random.default_rng(15)
    a, b = O.quantize(rng.random(16)), O.quantize(rng.random(16))
    k = 10
    c = O.Ctx(store=True, keygen=True, plan_k=k)
    c.distance("euclidean", a, b)
    bp = c.break_points()
    stream, sel = c.advice(), c.selectors()
    max_rows = (1 << k) - 9
    assert len(bp) == math.ceil(len(stream) / max_rows) - 1 or len(bp) == math.ceil(len(stream) / (max_rows - 1)) - 1
    cols = O.layout_columns(stream, bp, k, len(bp) + 1)
    assert cols.shape[0] == len(bp) + 1
    # every gate of the stream is intact inside one column; boundary cell duplicated at row 0
    pos = 0
    for ci in range(cols.shape[0]):
        rows = int(bp[ci]) + 1 if ci < len(bp) else len(stream) - pos
        assert np.array_equal(cols[ci, :rows], stream[pos:pos + rows])
        for r in np.nonzero(sel[pos:pos + rows])[0]:
            if not (ci < len(bp) and r == rows - 1):  # a gate starting on the break row restarts at row 0 of next col
                assert r + 3 < rows
        pos += rows - 1 if ci < len(bp) else rows
    lk = O.layout_lookup(c.lookup(), k, 8)
    assert lk.shape[0] == math.ceil(c.n_lookup / max_rows)


def test_the_rest_of_fixed_point_instructions_vs_python_and_f64(O, PY):
    """sign, clip, qmod, qsin, qcos, qtan, qsinh, qcosh, qtanh (fixed_point.rs:558-629, 817-916, 383-417; examples/fixed_point.rs reaches
    qsin): the C restatement against the independent Python one, results against f64 (the example prints the error, it holds no tolerance:
    1e-6 relative as everywhere), every gate row of the emitted cells satisfied"""
    import math
    fp = PY.FixedPoint(48)
    rng = np.random.default_rng(77)
    f64 = dict(qsin=math.sin, qcos=math.cos, qtan=math.tan, qsinh=math.sinh, qcosh=math.cosh, qtanh=math.tanh)
    xs = [float(v) for v in rng.uniform(-12.0, 12.0, 10)] + [0.25 * math.pi, 1.0, -1.0, 3.0, -3.5, 6.2, 0.001]
    c = O.Ctx(store=True, keygen=True)
    for x in xs:
        q = O.quantize([x])[0]
        i = O.fr_to_ints(q)[0]
        for name in ("sign", "clip", "qsin", "qcos", "qtan", "qsinh", "qcosh", "qtanh"):
            got = O.fr_to_ints(c.op(name, q))[0]
            assert got == getattr(fp, name)(i), (name, x)
            if name in f64:
                want = f64[name](x)
                assert abs(float(O.dequantize(O.fr_from_ints([got]))[0]) - want) <= 1e-6 * max(abs(want), 1.0), (name, x)
        assert O.fr_to_ints(c.op("clip", q))[0] == i                      # inside the range clip is the identity
        assert O.fr_to_ints(c.op("sign", q))[0] == (O.R_MOD - 1 if x < 0 else 1)
        for m in (2.0, 0.75, 2 * math.pi):
            qm = O.quantize([m])[0]
            got = O.fr_to_ints(c.op("qmod", q, qm))[0]
            assert got == fp.qmod(i, O.fr_to_ints(qm)[0])
            assert abs(float(O.dequantize(O.fr_from_ints([got]))[0]) - (x % m)) <= 1e-6 * m, (x, m)       # Python's % is the floored one too
    big = (1 << 96) + 12345                                              # outside the chip's range: clip brings it back modulo 2^(2P)
    assert O.fr_to_ints(c.op("clip", O.fr_from_ints([big])[0]))[0] == 12345 == fp.clip(big)
    neg_big = O.R_MOD - big                                              # ... and keeps the sign
    assert O.fr_to_ints(c.op("clip", O.fr_from_ints([neg_big])[0]))[0] == O.R_MOD - 12345 == fp.clip(neg_big)
    assert c.err == 0 and c.check_gates(13) == 0
