"""GPU parity of the batched MSM (C ABI: vdb_srs_load / vdb_msm*) against the CPU oracle
(halo2 `best_multiexp` restatement) — exact group elements, canonical affine, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init()
    return a


def witness_like(O, rng, n):
    """scalar mix of real witness columns: zeros, ones, bits, limbs, powers of two, negatives, products"""
    kinds = rng.integers(0, 10, size=n)
    vals = []
    for kd in kinds:
        if kd < 3:
            vals.append(0)
        elif kd < 5:
            vals.append(1)
        elif kd == 5:
            vals.append(int(rng.integers(0, 1 << 15)))
        elif kd == 6:
            vals.append(1 << int(rng.integers(0, 200)))
        elif kd == 7:
            vals.append(R - int(rng.integers(1, 1 << 62)))
        elif kd == 8:
            vals.append(int(rng.integers(0, 1 << 62)) << int(rng.integers(0, 60)))
        else:
            vals.append(int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) % R)
    return O.fr_from_ints(vals)


@pytest.mark.parametrize("k", [4, 8, 10])
def test_msm_random_and_edges(api, O, k):
    rng = np.random.default_rng(200 + k)
    n = 1 << k
    g, gl = O.srs_from_tau(k, 0xABCDEF12345 + k)
    srs = api.Srs(k, g, gl)
    cols = O.random_fr(rng, 4 * n).reshape(4, n, 4)
    cols[1] = witness_like(O, rng, n)
    cols[2] = 0                                   # all-zero column -> identity (0,0)
    cols[3] = O.fr_from_ints([R - 1] * n)         # all -1
    cols[0, :6] = O.fr_from_ints([0, 1, R - 1, (R - 1) // 2, (R + 1) // 2, 1 << 253])
    for basis, bases in ((1, gl), (0, g)):
        got = api.msm_batch(srs, cols, basis=basis)
        want = O.msm_batch(cols, bases, threads=4)
        assert np.array_equal(got, want), (k, basis)
    assert not got[2].any()
    # single-column entry point
    assert np.array_equal(api.msm(srs, cols[1], basis=1), O.msm(cols[1], gl))
    srs.free()


@pytest.mark.parametrize("k", [1, 2, 3, 5, 6, 12])
def test_msm_small_and_ragged(api, O, k):
    """tiny SRS sizes, shorter-than-SRS columns (n < 2^k is allowed), one column, and a wide batch of short columns (the
    range length is chosen per batch)"""
    rng = np.random.default_rng(900 + k)
    n_full = 1 << k
    g, _ = O.srs_from_tau(k, 0x51EDE + k)
    srs = api.Srs(k, g, None)
    for n_cols, n in ((1, n_full), (3, max(1, n_full - 1)), (2, max(1, n_full // 2)), (65 if k <= 6 else 5, n_full)):
        cols = np.stack([witness_like(O, rng, n) if c % 2 else O.random_fr(rng, n) for c in range(n_cols)])
        got = api.msm_batch(srs, cols, basis=0)
        assert np.array_equal(got, O.msm_batch(cols, g[:n], threads=4)), (k, n_cols, n)
    srs.free()


def test_msm_degenerate_bases(api, O):
    """tau = 2 makes many table points coincide (doubling / cancellation paths); identity bases too"""
    k = 8
    n = 1 << k
    g, gl = O.srs_from_tau(k, 2)
    g[5] = 0
    g[17] = g[3]
    g[18, :4] = g[3, :4]                      # -g[3]
    neg_y = (O.Q_MOD - O.fq_to_ints(g[3, 4:].reshape(1, 4))[0]) % O.Q_MOD
    g[18, 4:] = O.fq_from_ints([neg_y])[0]
    srs = api.Srs(k, g, None)
    rng = np.random.default_rng(300)
    cols = np.stack([witness_like(O, rng, n), O.random_fr(rng, n), O.fr_from_ints([1] * n), O.fr_from_ints([2] * n)])
    got = api.msm_batch(srs, cols, basis=0)
    assert np.array_equal(got, O.msm_batch(cols, g, threads=4))
    srs.free()


def test_msm_accumulator_exceptional_cases(api, O):
    """buckets that hold exactly {P, P}, {P, -P}, {P, P, -P}, {P, -P, P}: the accumulator's doubling and cancellation
    paths (k_msm_accum works in a lazy nine-limb form and detects them on a product), each window of the scalar"""
    k = 8
    n = 1 << k
    g, _ = O.srs_from_tau(k, 0x77AA55)
    g[17] = g[3]
    g[18, :4] = g[3, :4]
    neg_y = (O.Q_MOD - O.fq_to_ints(g[3, 4:].reshape(1, 4))[0]) % O.Q_MOD
    g[18, 4:] = O.fq_from_ints([neg_y])[0]
    g[40] = g[3]
    srs = api.Srs(k, g, None)
    rng = np.random.default_rng(77)
    s = int(O.fr_to_ints(O.random_fr(rng, 1))[0])
    cols = []
    for idx in ([3, 17], [3, 18], [3, 17, 18], [3, 18, 40], [3, 17, 40], [17, 18]):
        v = [0] * n
        for i in idx:
            v[i] = s
        cols.append(O.fr_from_ints(v))
    for small in (1, 5):                       # single-window scalars as well
        v = [0] * n
        v[3] = v[17] = small
        cols.append(O.fr_from_ints(v))
    cols = np.stack(cols)
    got = api.msm_batch(srs, cols, basis=0)
    assert np.array_equal(got, O.msm_batch(cols, g, threads=4))
    assert not got[1].any() and not got[5].any()          # P - P = identity, encoded (0, 0)
    srs.free()


def test_msm_linearity_k14(api, O):
    """size-independent property at a larger size: MSM(a + b) == MSM(a) + MSM(b) and against the
    closed form when the discrete logs of the bases are known."""
    k = 14
    n = 1 << k
    rng = np.random.default_rng(400)
    hs = [int(x) for x in rng.integers(1, 1 << 62, size=n)]
    bases = O.g1_mul_generator(hs)
    srs = api.Srs(k, None, bases)
    a = witness_like(O, rng, n)
    b = O.random_fr(rng, n)
    cols = np.stack([a, b, O.fr_add(a, b)])
    got = api.msm_batch(srs, cols, basis=1)
    ai, bi = O.fr_to_ints(a), O.fr_to_ints(b)
    ka = sum(x * h for x, h in zip(ai, hs)) % R
    kb = sum(x * h for x, h in zip(bi, hs)) % R
    want = O.g1_mul_generator([ka, kb, (ka + kb) % R])
    assert np.array_equal(got, want)
    srs.free()


def test_srs_setup_unsafe(api, O):
    k, tau = 6, 0x1234567
    g, gl = api.srs_setup_unsafe(k, O.fr_from_ints([tau])[0])
    wg, wgl = O.srs_from_tau(k, tau)
    assert np.array_equal(g, wg) and np.array_equal(gl, wgl)


@pytest.mark.parametrize("world", [2, 8])
def test_point_sharded_msm_combines_to_the_whole_commitment(O, world):
    """SURVEY §8(e), the alternative partition: the rows of every column split over `world` ranks (emulated one after the
    other), each committing its slice against the matching slice of the bases; the partial commitments added per column
    (vdb_g1_sum) are the whole commitments.  Three columns (fewer than ranks at world = 8: the case the partition is for),
    one of them all zero (identity partials)."""
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.pipeline import combine_partials, point_shard, point_sharded_partials
    api.init(0)
    k, n_cols = 10, 3
    rows = 1 << k
    rng = np.random.default_rng(88 + world)
    _, gl = O.srs_from_tau(k, 0xFACE)
    cols = O.random_fr(rng, n_cols * rows).reshape(n_cols, rows, 4)
    cols[1] = 0
    cols[2, : rows // 2] = 0            # ranks whose slice is all zero return the identity
    want = O.msm_batch(cols, gl)
    d = api.DeviceBuffer(cols.nbytes)
    d.upload(cols)
    parts = []
    for rank in range(world):
        lo, hi = point_shard(rows, rank, world)
        srs = api.Srs(k - int(np.log2(world)), None, gl[lo:hi])
        parts.append(point_sharded_partials(srs, d, n_cols, rows, lo, hi))
        srs.free()
    d.free()
    parts = np.stack(parts)
    assert not parts[:, 1].any() and not parts[0, 2].any()
    got = combine_partials(parts)
    assert np.array_equal(got, want) and not got[1].any()
    # the sum is order independent
    assert np.array_equal(combine_partials(parts[::-1]), want)


def test_g1_sum_exceptional_cases(O, PY):
    """vdb_g1_sum on the cases a generic addition formula gets wrong: equal summands (doubling), opposite summands (identity),
    identities among the summands, one summand only"""
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.pipeline import combine_partials
    api.init(0)
    P5, P7 = O.g1_mul_generator([5, 7])
    neg = lambda p: O.fq_from_ints([O.fq_to_ints(p.reshape(2, 4))[0], (-O.fq_to_ints(p.reshape(2, 4))[1]) % PY.Q]).reshape(8)
    zero = np.zeros(8, dtype=np.uint64)
    parts = np.stack([np.stack([P5, P5, zero, P5, zero]),
                      np.stack([P5, neg(P5), zero, zero, P7]),
                      np.stack([P5, zero, zero, P7, neg(P7)])])            # (3 ranks, 5 columns)
    got = combine_partials(parts)
    want = O.g1_mul_generator([15, 0, 0, 12, 0])
    assert np.array_equal(got, want)
    assert np.array_equal(combine_partials(parts[:1]), parts[0])


def test_multi_device_lifecycle_and_batches(api, O):
    """SURVEY 8(b) b0 behind the ABI: vdb_init_devices binds per-device contexts in one process, a host thread works on the
    device it selected, the *_multi entry points cut a batch of columns into one block per bound device (one host thread
    each) and return the commitments D2H.  This box has one GPU, so the blocks are one block — the same code path an 8-GPU
    caller runs with eight —, driven once from the main thread and once from a second host thread; an srs handle is refused
    on a device it does not belong to; shutdown / init cycles leave no stale per-device state (witness tables, twiddles)."""
    import threading
    rng = np.random.default_rng(77)
    k, n = 9, 512
    g, gl = O.srs_from_tau(k, 0x51DE)
    cols = O.random_fr(rng, 5 * n).reshape(5, n, 4)
    cols[1] = witness_like(O, rng, n)
    want = O.msm_batch(cols, gl, threads=4)
    qa = O.quantize(rng.uniform(-2, 2, (2, 5)))
    want_dist = api.wit_distance("euclidean", qa[:1], qa[1:], L=10)["stream"]
    try:
        api.init_devices(1)
        assert api.devices_bound() == 1 and api.current_device() == 0
        with pytest.raises(api.VdbError):
            api.set_device(1)                      # not bound
        srs = api.SrsAll(k, g, gl)
        assert srs.devices() == [0]
        assert np.array_equal(srs.msm_batch(cols), want)
        out = {}

        def worker():
            api.set_device(0)
            out["msm"] = srs.msm_batch(cols)
            out["ntt"] = api.ntt_batch_multi(cols, api.root_of_unity(k))
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert np.array_equal(out["msm"], want)
        assert np.array_equal(out["ntt"], O.ntt_batch(cols, O.root_of_unity(k), threads=2))
        srs.free()
        # cycle: everything cached on the device (gadget tables, twiddles, Poseidon spec) is rebuilt after a shutdown
        api.shutdown()
        import ctypes
        from halo2_vectordb_amd import _lib
        buf = (ctypes.c_uint64 * 4)()
        assert _lib.load().vdb_fr_mul(buf, buf, buf, ctypes.c_size_t(1)) == -1      # VDB_ERR_NOT_INIT: nothing is bound any more
        api.init(0)
        assert np.array_equal(api.wit_distance("euclidean", qa[:1], qa[1:], L=10)["stream"], want_dist)
        assert np.array_equal(api.ntt_batch(cols, api.root_of_unity(k)), out["ntt"])
        assert np.array_equal(api.poseidon_hash_many(cols[:1, :2])[0], O.poseidon_hash_many(cols[:1, :2])[0])
    finally:
        api.shutdown()
        api.init(0)


def test_two_device_contexts_in_one_process(api, O):
    """The n > 1 branch of the several-GPUs-in-one-process code (csrc/multi.hip for_each_device, per-device contexts and handles)
    on a one-GPU box: the test-only switch VDB_TEST_ALIAS_DEVICES=2 binds the card as two logical devices with separate contexts —
    streams, work space, twiddle tables, srs handles.  vdb_srs_load_all gives one handle per context; the *_multi entry points cut
    the batch into two blocks driven by two host threads; a handle used on the other context is refused with VDB_ERR_ARG; two host
    threads running the _dev pipeline (upload, commit, lagrange_to_coeff) concurrently on their own contexts give the
    single-context result."""
    import ctypes
    import os
    import threading
    from halo2_vectordb_amd._lib import check
    rng = np.random.default_rng(78)
    k, n = 10, 1024
    g, gl = O.srs_from_tau(k, 0x51DF)
    cols = O.random_fr(rng, 7 * n).reshape(7, n, 4)
    cols[2] = witness_like(O, rng, n)
    want = O.msm_batch(cols, gl, threads=4)
    lib = api._lib.init()
    os.environ["VDB_TEST_ALIAS_DEVICES"] = "2"
    try:
        api.init_devices(2)
        assert api.devices_bound() == 2 and api.current_device() == 0
        with pytest.raises(api.VdbError):
            api.set_device(2)                      # not bound
        srs = api.SrsAll(k, g, gl)
        assert srs.devices() == [0, 1]
        assert np.array_equal(srs.msm_batch(cols), want)                               # blocks [0, 3) and [3, 7) on two contexts
        assert np.array_equal(api.ntt_batch_multi(cols, api.root_of_unity(k)), O.ntt_batch(cols, O.root_of_unity(k), threads=2))
        # a handle belongs to its context
        out1 = np.zeros((1, 8), dtype=np.uint64)
        ptrs = (ctypes.c_void_p * 1)(cols[0].ctypes.data)
        api.set_device(1)
        assert lib.vdb_msm_batch(srs.handles[0], 1, ptrs, ctypes.c_size_t(1), ctypes.c_size_t(n), api._p(out1)) == -3      # VDB_ERR_ARG
        assert b"another device" in lib.vdb_last_error()
        assert lib.vdb_msm_batch(srs.handles[1], 1, ptrs, ctypes.c_size_t(1), ctypes.c_size_t(n), api._p(out1)) == 0 and np.array_equal(out1[0], want[0])
        api.set_device(0)
        # two host threads, each on its own context, the device-resident pipeline concurrently
        res, errs = {}, []

        def worker(dev, lo, hi):
            try:
                api.set_device(dev)
                m = hi - lo
                for rep in range(3):
                    buf = api.DeviceBuffer(m * n * 32)
                    buf.upload(np.ascontiguousarray(cols[lo:hi]))
                    com = np.zeros((m, 8), dtype=np.uint64)
                    check(lib.vdb_msm_batch_dev(srs.handles[dev], 1, buf.ptr, ctypes.c_size_t(m), ctypes.c_size_t(n), api._p(com)))
                    check(lib.vdb_lagrange_to_coeff_dev(buf.ptr, ctypes.c_size_t(m), ctypes.c_uint32(k)))
                    res[(dev, rep)] = (com, buf.download((m, n, 4)))
                    buf.free()
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))
        threads = [threading.Thread(target=worker, args=(0, 0, 4)), threading.Thread(target=worker, args=(1, 4, 7))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs, errs
        srs.free()
        api.shutdown()
        del os.environ["VDB_TEST_ALIAS_DEVICES"]
        # the single-context result
        api.init(0)
        single = api.Srs(k, g, gl)
        buf = api.DeviceBuffer(7 * n * 32)
        buf.upload(cols)
        com = np.zeros((7, 8), dtype=np.uint64)
        check(lib.vdb_msm_batch_dev(single.h, 1, buf.ptr, ctypes.c_size_t(7), ctypes.c_size_t(n), api._p(com)))
        check(lib.vdb_lagrange_to_coeff_dev(buf.ptr, ctypes.c_size_t(7), ctypes.c_uint32(k)))
        coeff = buf.download((7, n, 4))
        buf.free()
        single.free()
        assert np.array_equal(com, want)
        for rep in range(3):
            assert np.array_equal(np.concatenate([res[(0, rep)][0], res[(1, rep)][0]]), com)
            assert np.array_equal(np.concatenate([res[(0, rep)][1], res[(1, rep)][1]]), coeff)
        with pytest.raises(api.VdbError):
            api.init_devices(2)                                                      # without the switch the box has one device
    finally:
        os.environ.pop("VDB_TEST_ALIAS_DEVICES", None)
        api.shutdown()
        api.init(0)


@pytest.mark.parametrize("window_bits", [13, 14])
def test_msm_wide_windows(api, O, window_bits):
    """the windows chosen for columns of full-width scalars (product columns, Poseidon states; api.Srs(window_bits=14)): dense
    columns, a witness-like mix in which short and long scalars share wavefronts, edge values, ragged length; against the oracle"""
    rng = np.random.default_rng(300 + window_bits)
    k = 11
    n = 1 << k
    g, gl = O.srs_from_tau(k, 0xD161 + window_bits)
    srs = api.Srs(k, g, gl, window_bits=window_bits)
    assert srs.info()[1] == window_bits
    cols = O.random_fr(rng, 6 * n).reshape(6, n, 4)
    cols[1] = witness_like(O, rng, n)
    cols[2] = 0
    cols[3] = O.fr_from_ints([R - 1] * n)
    cols[4, ::3] = O.fr_from_ints([int(v) for v in rng.integers(0, 1 << 15, size=len(range(0, n, 3)))])     # short among long, lane by lane
    cols[0, :8] = O.fr_from_ints([0, 1, R - 1, (R - 1) // 2, (R + 1) // 2, 1 << 253, (1 << 42) - 1, 1 << 41])
    for basis, bases in ((1, gl), (0, g)):
        assert np.array_equal(api.msm_batch(srs, cols, basis=basis), O.msm_batch(cols, bases, threads=4)), basis
    short = cols[:, : n - 37]
    assert np.array_equal(api.msm_batch(srs, short, basis=0), O.msm_batch(short, g[: n - 37], threads=4))
    srs.free()


def test_msm_sweep_of_window_sizes_and_digit_boundaries(api, O):
    """every window size the table builder accepts (2 .. 14 bits) at random SRS sizes, batch widths and ragged lengths; the scalars
    include what sits on the edges of the signed-digit recoding: 2^(c j) - 1 and 2^(c j - 1) (a digit that carries / the largest digit
    that does not), their negatives, the fold point (r - 1) / 2, and the short / long classification's limit 2^32"""
    rng = np.random.default_rng(20261005)
    for case in range(26):
        c = 2 + case % 13
        k = int(rng.integers(3, 12))
        n_full = 1 << k
        n = n_full if case % 3 else int(rng.integers(1, n_full + 1))
        n_cols = int(rng.integers(1, 9))
        g, gl = O.srs_from_tau(k, 0xF00D + case)
        srs = api.Srs(k, g, gl, window_bits=c)
        assert srs.info()[1] == c
        edges = [0, 1, R - 1, (R - 1) // 2, (R + 1) // 2, (1 << 32) - 1, 1 << 32, R - (1 << 32), (1 << 253)]
        for j in range(1, 254 // c + 1):
            edges += [(1 << (c * j)) - 1, 1 << (c * j - 1), (1 << (c * j - 1)) + 1, R - (1 << (c * j - 1)), R - (1 << (c * j)) + 1]
        edges = [int(e) % R for e in edges]
        cols = []
        for col in range(n_cols):
            kind = (case + col) % 3
            v = O.random_fr(rng, n) if kind == 0 else witness_like(O, rng, n)
            if kind == 2:
                pick = [edges[int(i)] for i in rng.integers(0, len(edges), size=n)]
                v = O.fr_from_ints(pick)
            cols.append(v)
        cols = np.stack(cols)
        basis = case % 2
        bases = (g, gl)[basis][:n]
        got = api.msm_batch(srs, cols, basis=basis)
        assert np.array_equal(got, O.msm_batch(cols, bases, threads=4)), (case, c, k, n, n_cols, basis)
        srs.free()
