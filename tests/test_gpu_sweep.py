"""A seeded sweep over shapes the fixed cases of tests/test_gpu_witness.py do not name: every gadget entry point at random sizes, all
four distances, both precisions the reference's examples use, lookup widths 8 .. 15 — advice cells, lookup cells, gate flags and
results against the oracle, bit for bit; then the same traces stored through random rank windows (vdb_wit_set_window): every cell
inside the window equal, every cell outside it untouched.  Where the oracle reports a domain error (the reference would panic: an
empty cluster's division) the library must refuse with VDB_ERR_DOMAIN instead of returning cells."""
import ctypes

import numpy as np
import pytest

from test_gpu_witness import assert_streams

pytestmark = pytest.mark.gpu

METRICS = ("euclidean", "cosine", "manhattan", "hamming")


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init()
    return a


def _vectors(rng, n, dim, metric):
    v = rng.uniform(-2.5, 2.5, size=(n, dim))
    if metric == "hamming":                       # equal elements have to exist for the count to mean anything
        v = rng.integers(0, 3, size=(n, dim)).astype(np.float64) * 0.5
    if metric == "cosine":                        # no zero vector: the cosine of one is a division by zero
        v[np.abs(v).sum(axis=1) == 0, 0] = 1.0
    return v


def test_random_shapes_of_every_gadget_against_the_oracle(api, O):
    rng = np.random.default_rng(20261005)
    refused = 0
    for case in range(36):
        metric = METRICS[case % 4]
        P = 48 if case % 5 else 32
        L = int(rng.integers(8, 16))
        kind = ("distance", "nearest", "kmeans")[case % 3]
        dim = int(rng.choice([1, 2, 3, 5, 8, 17, 64, 65, 129]))
        c = O.Ctx(store=True, keygen=True)
        tag = f"case {case}: {kind} {metric} P={P} L={L} dim={dim}"
        if kind == "distance":
            n = int(rng.integers(1, 70))
            a, b = _vectors(rng, n, dim, metric), _vectors(rng, n, dim, metric)
            qa, qb = O.quantize(a, P), O.quantize(b, P)
            want = np.stack([c.distance(metric, x, y, P=P, L=L) for x, y in zip(qa, qb)])
            call = lambda: api.wit_distance(metric, qa, qb, P=P, L=L, selectors=True)
            results = lambda got: np.array_equal(got["result"], want)
        elif kind == "nearest":
            n = int(rng.integers(1, 40))
            db, q = _vectors(rng, n, dim, metric), _vectors(rng, 1, dim, metric)[0]
            qq, qdb = O.quantize(q, P), O.quantize(db, P)
            ind, res = c.nearest_vector(metric, qq, qdb, P=P, L=L)
            call = lambda: api.wit_nearest(metric, qq, qdb, P=P, L=L, selectors=True)
            results = lambda got: np.array_equal(got["indicator"], ind) and np.array_equal(got["result"], res)
        else:
            n = int(rng.integers(3, 40))
            K, I = int(rng.integers(1, min(n, 5))), int(rng.integers(1, 4))
            tag += f" n={n} K={K} I={I}"
            qv = O.quantize(_vectors(rng, n, dim, metric) + (0.0 if metric == "hamming" else 3.0), P)
            cent, ind = c.kmeans(metric, qv, K, I, P=P, L=L)
            call = lambda: api.wit_kmeans(metric, qv, K, I, P=P, L=L, selectors=True)
            results = lambda got: np.array_equal(got["centroids"], cent) and np.array_equal(got["indicators"], ind)
        if c.err:
            with pytest.raises(api.VdbError) as e:
                call()
            assert e.value.code == -5, tag
            refused += 1
            continue
        got = call()
        assert results(got), tag
        try:
            assert_streams(got, c)
        except AssertionError as e:
            raise AssertionError(tag + ": " + str(e)) from e
    assert refused <= 12          # the sweep is not allowed to degenerate into error cases


def test_random_rank_windows_store_their_cells_and_nothing_else(api, O):
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    rng = np.random.default_rng(777)
    POISON = np.uint64(0xA5A5A5A5A5A5A5A5)
    done = 0
    for case in range(14):
        metric = METRICS[case % 4]
        P, L = 48, int(rng.integers(9, 14))
        n, dim = int(rng.integers(6, 30)), int(rng.choice([3, 8, 33, 70]))
        K, I = int(rng.integers(2, 4)), int(rng.integers(1, 3))
        qv = O.quantize(_vectors(rng, n, dim, metric) + (0.0 if metric == "hamming" else 3.0), P)
        c = O.Ctx(store=True)
        cent, ind = c.kmeans(metric, qv, K, I, P=P, L=L)
        if c.err:
            continue
        adv, lk = c.advice(), c.lookup()
        lo, hi = sorted(int(x) for x in rng.integers(0, len(adv) + 1, 2))
        llo, lhi = sorted(int(x) for x in rng.integers(0, len(lk) + 1, 2))
        bufs = [api.DeviceBuffer(max(x, 32)) for x in (qv.nbytes, adv.nbytes, lk.nbytes, cent.nbytes, ind.nbytes)]
        d_vec, d_adv, d_lk, d_cent, d_ind = bufs
        try:
            d_vec.upload(qv)
            check(lib.vdb_memset_dev(d_adv.ptr, 0xA5, ctypes.c_size_t(adv.nbytes)))
            check(lib.vdb_memset_dev(d_lk.ptr, 0xA5, ctypes.c_size_t(max(lk.nbytes, 32))))
            check(lib.vdb_wit_set_window(ctypes.c_uint64(lo), ctypes.c_uint64(hi), ctypes.c_uint64(llo), ctypes.c_uint64(lhi)))
            try:
                check(lib.vdb_wit_kmeans_dev(api.METRICS[metric], P, L, d_vec.ptr, n, dim, K, I, 0, d_adv.ptr, d_lk.ptr, None, d_cent.ptr, d_ind.ptr))
            finally:
                check(lib.vdb_wit_set_window(ctypes.c_uint64(0), ctypes.c_uint64(2 ** 64 - 1), ctypes.c_uint64(0), ctypes.c_uint64(2 ** 64 - 1)))
            g_adv, g_lk = d_adv.download(adv.shape), d_lk.download(lk.shape) if len(lk) else lk
            tag = f"case {case}: {metric} n={n} dim={dim} K={K} I={I} L={L} window [{lo}, {hi}) / [{llo}, {lhi})"
            assert np.array_equal(g_adv[lo:hi], adv[lo:hi]), tag
            assert (g_adv[:lo] == POISON).all() and (g_adv[hi:] == POISON).all(), tag
            if len(lk):
                # a lookup cell is stored by the rank that stores the advice cell it copies: inside both windows it must be there,
                # outside the lookup window it must not
                inside = g_lk[llo:lhi]
                stored = ~(inside == POISON).all(axis=1)
                assert np.array_equal(inside[stored], lk[llo:lhi][stored]), tag
                assert (g_lk[:llo] == POISON).all() and (g_lk[lhi:] == POISON).all(), tag
                if lo == 0 and hi == len(adv):
                    assert stored.all(), tag
            # every rank computes every value
            assert np.array_equal(d_cent.download(cent.shape), cent) and np.array_equal(d_ind.download(ind.shape), ind), tag
            done += 1
        finally:
            for b in bufs:
                b.free()
    assert done >= 8


def _windowed(api, lib, check, adv, lk, window, run):
    """runs `run(d_adv, d_lk)` on poisoned device streams under the rank window; -> the streams as left behind"""
    POISON = 0xA5
    lo, hi, llo, lhi = window
    d_adv, d_lk = api.DeviceBuffer(max(adv.nbytes, 32)), api.DeviceBuffer(max(lk.nbytes, 32))
    try:
        check(lib.vdb_memset_dev(d_adv.ptr, POISON, ctypes.c_size_t(max(adv.nbytes, 32))))
        check(lib.vdb_memset_dev(d_lk.ptr, POISON, ctypes.c_size_t(max(lk.nbytes, 32))))
        check(lib.vdb_wit_set_window(ctypes.c_uint64(lo), ctypes.c_uint64(hi), ctypes.c_uint64(llo), ctypes.c_uint64(lhi)))
        try:
            run(d_adv, d_lk)
        finally:
            check(lib.vdb_wit_set_window(ctypes.c_uint64(0), ctypes.c_uint64(2 ** 64 - 1), ctypes.c_uint64(0), ctypes.c_uint64(2 ** 64 - 1)))
        return d_adv.download(adv.shape), (d_lk.download(lk.shape) if len(lk) else lk)
    finally:
        d_adv.free()
        d_lk.free()


def _check_window(adv, lk, g_adv, g_lk, window, tag):
    POISON = np.uint64(0xA5A5A5A5A5A5A5A5)
    lo, hi, llo, lhi = window
    assert np.array_equal(g_adv[lo:hi], adv[lo:hi]), tag
    assert (g_adv[:lo] == POISON).all() and (g_adv[hi:] == POISON).all(), tag
    if len(lk):
        inside = g_lk[llo:lhi]
        stored = ~(inside == POISON).all(axis=1)
        assert np.array_equal(inside[stored], lk[llo:lhi][stored]), tag
        assert (g_lk[:llo] == POISON).all() and (g_lk[lhi:] == POISON).all(), tag
        if lo == 0 and hi == len(adv):
            assert stored.all(), tag


def test_rank_windows_of_the_other_device_entry_points(api, O):
    """vdb_wit_distance_dev, vdb_wit_nearest_dev and vdb_wit_merkle_dev under random rank windows (the k-means entry point: above)"""
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    rng = np.random.default_rng(4711)
    P = 48
    for case in range(12):
        metric = METRICS[case % 4]
        L = int(rng.integers(9, 14))
        n, dim = int(rng.integers(2, 24)), int(rng.choice([2, 5, 16, 67]))
        c = O.Ctx(store=True)
        up = []

        def dev(a):
            b = api.DeviceBuffer(max(a.nbytes, 32))
            b.upload(np.ascontiguousarray(a))
            up.append(b)
            return b
        try:
            kind = case % 3
            if kind == 0:
                qa, qb = O.quantize(_vectors(rng, n, dim, metric), P), O.quantize(_vectors(rng, n, dim, metric), P)
                want = np.stack([c.distance(metric, x, y, P=P, L=L) for x, y in zip(qa, qb)])
                d_a, d_b, d_res = dev(qa), dev(qb), dev(np.zeros_like(want))
                run = lambda d_adv, d_lk: check(lib.vdb_wit_distance_dev(api.METRICS[metric], P, L, d_a.ptr, d_b.ptr, n, dim, d_adv.ptr, d_lk.ptr, None, d_res.ptr))
                values = lambda: np.array_equal(d_res.download(want.shape), want)
            elif kind == 1:
                q, db = O.quantize(_vectors(rng, 1, dim, metric)[0], P), O.quantize(_vectors(rng, n, dim, metric), P)
                ind, res = c.nearest_vector(metric, q, db, P=P, L=L)
                d_q, d_db, d_ind, d_res = dev(q), dev(db), dev(np.zeros_like(ind)), dev(np.zeros_like(res))
                run = lambda d_adv, d_lk: check(lib.vdb_wit_nearest_dev(api.METRICS[metric], P, L, d_q.ptr, d_db.ptr, n, dim, d_adv.ptr, d_lk.ptr, None, d_ind.ptr, d_res.ptr))
                values = lambda: np.array_equal(d_ind.download(ind.shape), ind) and np.array_equal(d_res.download(res.shape), res)
            else:
                v = O.quantize(rng.random((n, dim)), P)
                root = c.merkle_commitment(v)
                d_v, d_root = dev(v), dev(np.zeros_like(root))
                run = lambda d_adv, d_lk: check(lib.vdb_wit_merkle_dev(d_v.ptr, n, dim, 0, d_adv.ptr, None, d_root.ptr))
                values = lambda: np.array_equal(d_root.download(root.shape), root)
            assert c.err == 0
            adv, lk = c.advice(), c.lookup()
            lo, hi = sorted(int(x) for x in rng.integers(0, len(adv) + 1, 2))
            llo, lhi = sorted(int(x) for x in rng.integers(0, len(lk) + 1, 2)) if len(lk) else (0, 0)
            for window in ((lo, hi, llo, lhi), (0, len(adv), 0, len(lk))):
                tag = f"case {case}: kind {kind} {metric} n={n} dim={dim} L={L} window {window}"
                g_adv, g_lk = _windowed(api, lib, check, adv, lk, window, run)
                _check_window(adv, lk, g_adv, g_lk, window, tag)
                assert values(), tag
        finally:
            for b in up:
                b.free()


def test_break_points_and_columns_of_random_gadget_sequences(api, O):
    """halo2-base's column layout (GateThreadBuilder::assign_all: a column ends where the next gate would not fit, the cell two gates
    share is assigned again at the top of the next) on streams no fixed gadget produces: random sequences of FixedPointChip operations
    and bare witness cells at small column heights, so that breaks fall after every kind of cell — break points, advice columns and
    lookup columns against the oracle's Context"""
    rng = np.random.default_rng(60606)
    ops1 = ("neg", "qabs", "is_neg", "signed_div_scale")
    ops2 = ("qadd", "qsub", "qmul", "qmin", "qmax", "qdiv")
    for case in range(14):
        k = int(rng.integers(6, 11))
        L = int(rng.integers(4, k))                       # the lookup table must fit the column
        c = O.Ctx(store=True, keygen=True, plan_k=k)
        vals = [O.quantize(np.array([x]))[0] for x in rng.uniform(-4, 4, 6)]
        for _ in range(int(rng.integers(3, 40))):
            r = rng.random()
            if r < 0.2:
                c.assign_witnesses(np.stack([vals[int(i)] for i in rng.integers(0, len(vals), int(rng.integers(1, 9)))]))
            elif r < 0.5:
                vals.append(c.op(ops1[int(rng.integers(0, len(ops1)))], vals[int(rng.integers(0, len(vals)))], L=L))
            else:
                a, b = vals[int(rng.integers(0, len(vals)))], vals[int(rng.integers(0, len(vals)))]
                name = ops2[int(rng.integers(0, len(ops2)))]
                if name == "qdiv" and not O.fr_to_ints(b.reshape(1, 4))[0]:
                    name = "qmul"
                vals.append(c.op(name, a, b, L=L))
            vals = vals[-8:]
        assert c.err == 0
        adv, lk, flags = c.advice(), c.lookup(), c.selectors().astype(np.uint8)
        bp = api.layout_plan(flags, k)
        assert np.array_equal(bp, c.break_points()), (case, k, len(adv))
        cols, lcols = api.layout_columns(adv, bp, k, lookup=lk if len(lk) else None)
        assert np.array_equal(cols, O.layout_columns(adv, c.break_points(), k, len(bp) + 1)), (case, k)
        if len(lk):
            assert np.array_equal(lcols, O.layout_lookup(lk, k, lcols.shape[0])), (case, k)


def test_permutation_of_random_copy_forests(api, O):
    """keygen's cycle construction on the device (vdb_permutation_mapping_dev: pointer jumping over the copy forest, one sort) on copy
    structures no gadget produces: random forests of any depth over the cells of a small circuit, roots tied to random constants, lookup
    cells copying random cells, public rows naming random cells (some twice) — the sigma commitments against the CPU prover's
    construction (oracle/prover.py permutation_mapping: an explicit root search and a stable sort)"""
    from halo2_vectordb_amd import circuit_sym as CS
    from halo2_vectordb_amd.pipeline import DistancesHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import prover as PV
    TAU = 0x1234567890ABCDEF1234567
    rng = np.random.default_rng(90210)
    k, L = 9, 8
    hp = DistancesHotPath(dim=3, metrics=("manhattan", "hamming"), k=k, L=L, tau=TAU).setup()
    g, gl = O.srs_from_tau(k, TAU)
    d_flags = hp.keygen_flags()
    flags = d_flags.download((hp.n_cells,), dtype=np.uint8)
    d_flags.free()
    try:
        nc, nl = hp.n_cells, hp.n_lookup
        assert hp.n_adv_cols >= 3 and nl > 0
        for case in range(6):
            p_copy = (0.0, 0.3, 0.6, 0.9, 0.98, 0.5)[case]
            idx = np.arange(nc, dtype=np.int64)
            parent = np.where(rng.random(nc) < p_copy, (rng.random(nc) * (idx + 1)).astype(np.int64), idx)     # an earlier cell, or itself
            parent[0] = 0
            if case == 4:
                parent = np.maximum(idx - 1, 0)                               # one chain through every cell: the deepest forest there is
            consts = [int(x) for x in rng.integers(0, 1 << 40, int(rng.integers(1, 9)))]
            consts = list(dict.fromkeys(consts))
            const_idx = np.full(nc, -1, dtype=np.int64)
            roots = np.flatnonzero(parent == idx)
            tied = roots[rng.random(roots.size) < 0.2]
            const_idx[tied] = rng.integers(0, len(consts), tied.size)
            lookup_src = rng.integers(0, nc, nl).astype(np.int64)
            inst = [int(x) for x in rng.integers(0, nc, int(rng.integers(0, 7)))]
            if len(inst) > 2:
                inst[-1] = inst[0]                                            # two public rows naming one cell
            cm = CS.CopyMap(parent, const_idx, consts, np.zeros(nc, dtype=bool), (flags & 1).astype(bool), lookup_src)
            pr = ProverRounds(hp).keygen(circuit=cm, instance_cells=inst, check=False)
            try:
                cs = PV.Circuit(k, L, np.asarray(hp.bp), (flags & 1), nl, parent, const_idx, consts, lookup_src, inst)
                assert (cs.n_adv, cs.n_lk, cs.n_sets) == (pr.n_adv, pr.n_lk, pr.n_sets)
                pk = PV.keygen(cs, g, gl, threads=4)
                for name in ("sigma", "cst", "sel", "table"):
                    assert np.array_equal(pk.commits[name], pr.fixed[name].commits), (case, name)
            finally:
                pr.free()
    finally:
        hp.free()


def test_device_mock_prover_counts_what_a_host_recount_counts(api, O):
    """vdb_mock_check_dev on corrupted witnesses: a nearest_vector witness with one random cell (or lookup cell) replaced, again and
    again — the violation counts per kind (gate rows, copies, constants, lookup copies, cells outside the table) and the first offenders
    must be those of a recount in Python integers over the same arrays"""
    from halo2_vectordb_amd import circuit_sym as CS
    R = O.R_MOD
    rng = np.random.default_rng(1234321)
    n, dim, P, L = 5, 4, 48, 10
    cm, _ = CS.build_nearest("euclidean", n, dim, P, L, builder=None)
    v = rng.uniform(0.1, 2.0, size=(n + 1, dim))
    q, db = api.quantize([v[0]])[0], api.quantize(v[1:])
    got = api.wit_nearest("euclidean", q, db, P=P, L=L, selectors=True)
    stream = np.concatenate([q, db.reshape(-1, 4), got["stream"]])
    flags = np.concatenate([np.zeros((n + 1) * dim, dtype=np.uint8), got["flags"]])
    lookup = got["lookup"]
    nc, nl = stream.shape[0], lookup.shape[0]
    assert nc == cm.n_cells and nl == len(cm.lookup_src)
    table = O.fr_from_ints(cm.consts)
    consts = np.array(cm.consts, dtype=object)
    gates = np.flatnonzero(flags & 1)
    copies = np.flatnonzero(cm.copy_of != np.arange(nc))
    tied = np.flatnonzero(cm.const_idx >= 0)

    def recount(vals, lks):
        a, b, c, d = (vals[gates + i] for i in range(4))
        bad_g = gates[np.array([(int(w) + int(x) * int(y) - int(z)) % R != 0 for w, x, y, z in zip(a, b, c, d)], dtype=bool)]
        bad_c = copies[vals[copies] != vals[cm.copy_of[copies]]]
        bad_k = tied[vals[tied] != consts[cm.const_idx[tied]]]
        bad_l = np.flatnonzero(lks != vals[cm.lookup_src])
        out_t = np.flatnonzero(np.array([int(x) >= (1 << L) for x in lks], dtype=bool))
        return bad_g, bad_c, bad_k, bad_l, out_t

    bufs = {}

    def dev(name, a):
        a = np.ascontiguousarray(a)
        if name not in bufs:
            bufs[name] = api.DeviceBuffer(max(a.nbytes, 32))
        bufs[name].upload(a)
        return bufs[name].ptr
    try:
        p_flags, p_copy, p_src = dev("flags", flags), dev("copy", cm.copy_of), dev("src", cm.lookup_src)
        p_cidx, p_table = dev("cidx", cm.const_idx), dev("table", table)
        seen = set()
        for trial in range(40):
            s2, l2 = stream.copy(), lookup.copy()
            if trial == 0:
                pass                                                         # the honest witness
            elif trial % 5 == 4:
                j = int(rng.integers(0, nl))
                l2[j] = O.fr_from_ints([int(rng.integers(0, 1 << (L + 2)))])[0]
            else:
                i = int(rng.integers(0, nc))
                s2[i] = O.random_fr(rng, 1)[0] if trial % 2 else O.fr_add(s2[i].reshape(1, 4), O.fr_from_ints([1]))[0]
            vals, lks = np.array(O.fr_to_ints(s2), dtype=object), np.array(O.fr_to_ints(l2), dtype=object)
            bad_g, bad_c, bad_k, bad_l, out_t = recount(vals, lks)
            rep = api.mock_check_dev(dev("stream", s2), nc, p_flags, dev("lookup", l2), nl, L, p_copy, p_src, None, p_cidx, p_table, len(cm.consts))
            want = (len(bad_g), len(bad_c), len(bad_k), len(bad_l), len(out_t))
            assert (rep.gate_rows_violated, rep.copies_unequal, rep.constants_changed, rep.lookup_copies_unequal, rep.lookup_cells_out_of_table) == want, (trial, rep.as_dict(), want)
            for cnt, first, bad in ((rep.gate_rows_violated, rep.first_gate_row, bad_g), (rep.copies_unequal, rep.first_copy, bad_c),
                                    (rep.constants_changed, rep.first_constant, bad_k), (rep.lookup_copies_unequal, rep.first_lookup_copy, bad_l),
                                    (rep.lookup_cells_out_of_table, rep.first_lookup_cell, out_t)):
                if cnt:
                    assert first == int(bad[0]), (trial, rep.as_dict())
            seen.add(tuple(x > 0 for x in want))
            if trial == 0:
                assert want == (0, 0, 0, 0, 0)
        assert len(seen) >= 4            # the corruptions hit several kinds of constraint
    finally:
        for b in bufs.values():
            b.free()


def test_rank_windows_of_the_single_operation_entry_point(api, O):
    """vdb_wit_fp_op_dev under random rank windows: the cells inside, nothing outside, every result on every rank"""
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    rng = np.random.default_rng(5150)
    P, L, n = 48, 11, 37
    for name in ("qmul", "qdiv", "qsqrt", "qsin", "qtanh", "qmod"):
        x = rng.uniform(0.2, 6.0, n)
        y = rng.uniform(0.3, 5.0, n)
        qa, qb = O.quantize(x, P), O.quantize(y, P)
        binary = name in ("qmul", "qdiv", "qmod")
        c = O.Ctx(store=True)
        want = np.stack([c.op(name, qa[i], qb[i] if binary else None, P=P, L=L) for i in range(n)])
        adv, lk = c.advice(), c.lookup()
        bufs = [api.DeviceBuffer(qa.nbytes), api.DeviceBuffer(qb.nbytes), api.DeviceBuffer(want.nbytes)]
        try:
            bufs[0].upload(qa)
            bufs[1].upload(qb)
            lo, hi = sorted(int(v) for v in rng.integers(0, len(adv) + 1, 2))
            llo, lhi = sorted(int(v) for v in rng.integers(0, len(lk) + 1, 2))
            run = lambda d_adv, d_lk: check(lib.vdb_wit_fp_op_dev(api.FP_OPS[name], P, L, bufs[0].ptr, bufs[1].ptr if binary else None, n, d_adv.ptr, d_lk.ptr,
                                                                  None, bufs[2].ptr))
            for window in ((lo, hi, llo, lhi), (0, len(adv), 0, len(lk))):
                g_adv, g_lk = _windowed(api, lib, check, adv, lk, window, run)
                _check_window(adv, lk, g_adv, g_lk, window, (name, window))
                assert np.array_equal(bufs[2].download(want.shape), want), name
        finally:
            for b in bufs:
                b.free()
