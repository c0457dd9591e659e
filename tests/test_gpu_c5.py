"""BASELINE configs[4] — the demo pipeline (/root/reference/tests/demo/mod.rs:47-90, tests/demo_test.rs:59-88: kmeans::<2, 1>
index, nearest_vector queries, Merkle commitments) at its parameters k = 18, LOOKUP_BITS = 17, through the C ABI:

* the gadget cells at LOOKUP_BITS = 17 (kmeans<2,1>, nearest_vector over 128-dim SIFT-shaped vectors) equal the oracle's bit
  for bit, on a few hundred vectors so that the oracle finishes in seconds;
* the resident hot path at 2^18 rows (break points, laid-out columns, sampled commitments, an extended coset) equals the oracle;
* one column shard of the full-size database Merkle circuit (10,000 x 128 vectors, 1.54 G cells, 5,881 columns: the shard an
  8-GPU job gives one rank) ends in the oracle's root, holds the oracle's leaf traces and commits to the oracle's points.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
K18, L17, N_BLIND = 18, 17, 7


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init()
    return a


def sift(seed, n, dim=128):
    v = np.random.default_rng(seed).integers(0, 219, size=(n, dim)).astype(np.float64)
    v[v.sum(axis=1) == 0, 0] = 1
    return v


def check_hot_path_against_oracle(O, hp, c, n_ext=1):
    """break points, sampled laid-out columns, their commitments and one extended coset of a set-up hot path `hp` against the
    oracle context `c` that holds the same circuit"""
    assert hp.n_cells == len(c) and hp.n_lookup == c.n_lookup
    assert np.array_equal(hp.bp, c.break_points())
    com = hp.step().copy()
    ext0 = hp.d_ext.download((n_ext, 4 * hp.rows, 4))
    coeff0 = hp.d_cols.download((n_ext, hp.rows, 4))
    hp.relayout()
    pick = sorted({0, hp.n_adv_cols // 2, hp.n_adv_cols - 1, hp.n_adv_cols, hp.n_cols - 1} if hp.n_lk_cols else {0, hp.n_adv_cols // 2, hp.n_adv_cols - 1})
    cols = hp.download_columns(pick)
    adv, lk = c.advice(), c.lookup()
    starts = np.concatenate([[0], np.cumsum(hp.bp, dtype=np.uint64)]).astype(np.int64)
    used, max_rows = hp.rows - N_BLIND, hp.rows - 9
    for j, col in enumerate(pick):
        if col < hp.n_adv_cols:       # column c = stream[starts[c] : starts[c] + bp[c] + 1] (the boundary cell is shared with the next column)
            ln = int(hp.bp[col]) + 1 if col < len(hp.bp) else hp.n_cells - int(starts[col])
            want = adv[starts[col]: starts[col] + ln]
        else:
            lo = (col - hp.n_adv_cols) * max_rows
            want = lk[lo: lo + max_rows]
        assert np.array_equal(cols[j][: len(want)], want), f"column {col}"
        assert not cols[j][len(want): used].any()
    assert np.array_equal(com[pick], O.msm_batch(cols, hp.g_lagrange, threads=8)), "commitments differ from the oracle"
    first = hp.download_columns(list(range(n_ext)))
    wc, we = O.lde_batch(first, ext=2, threads=n_ext)
    assert np.array_equal(coeff0, wc) and np.array_equal(ext0, we), "lagrange_to_coeff / coeff_to_extended differ from the oracle"


def test_c5_kmeans_k18_lookup_bits_17(api, O):
    """kmeans::<2, 1> (tests/demo_test.rs:62-63) at LOOKUP_BITS = 17 over 160 x 128 SIFT-shaped vectors, then at 2^18 rows"""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    n = 160
    hp = KmeansHotPath(n=n, dim=128, K=2, I=1, k=K18, P=48, L=L17, seed=20260005, tau=0xC5).setup()
    qv = O.quantize(hp.vectors_f64)
    c = O.Ctx(store=True, keygen=True, plan_k=K18)
    c.assign_witnesses(qv)
    cent, ind = c.kmeans("euclidean", qv, 2, 1, P=48, L=L17)
    assert c.err == 0
    got = api.wit_kmeans("euclidean", qv, 2, 1, P=48, L=L17, selectors=True)
    assert np.array_equal(got["centroids"], cent) and np.array_equal(got["indicators"], ind)
    off = n * 128
    assert np.array_equal(got["stream"], c.advice()[off:]) and np.array_equal(got["lookup"], c.lookup())
    assert np.array_equal(got["selectors"], c.selectors()[off:] & 1)
    del got
    check_hot_path_against_oracle(O, hp, c)
    gc, gi = hp.results()
    assert np.array_equal(gc, cent) and np.array_equal(gi, ind)
    # the f64 k-means of the reference's tests (tests/vectordb/mod.rs:31-91) on the same vectors: exact cluster ids
    d = np.linalg.norm(hp.vectors_f64[:, None, :] - hp.vectors_f64[None, :2, :], axis=2)
    ids = api.dequantize(gi.reshape(-1, 4)).reshape(n, 2).argmax(axis=1)
    assert np.array_equal(ids, d.argmin(axis=1))
    hp.free()


def test_c5_nearest_k18_lookup_bits_17(api, O):
    """nearest_vector inside a cluster (tests/demo/mod.rs:82) at LOOKUP_BITS = 17 over 200 x 128 vectors + query, at 2^18 rows"""
    from halo2_vectordb_amd.pipeline import NearestHotPath
    n = 200
    hp = NearestHotPath(n=n, dim=128, k=K18, P=48, L=L17, seed=20260005, tau=0xC5).setup()
    qq, qdb = hp.qvec[0], hp.qvec[1:]
    c = O.Ctx(store=True, keygen=True, plan_k=K18)
    c.assign_witnesses(qq)
    c.assign_witnesses(qdb)
    ind, res = c.nearest_vector("euclidean", qq, qdb, P=48, L=L17)
    got = api.wit_nearest("euclidean", qq, qdb, P=48, L=L17, selectors=True)
    off = (n + 1) * 128
    assert np.array_equal(got["indicator"], ind) and np.array_equal(got["result"], res)
    assert np.array_equal(got["stream"], c.advice()[off:]) and np.array_equal(got["lookup"], c.lookup())
    assert np.array_equal(got["selectors"], c.selectors()[off:] & 1)
    del got
    check_hot_path_against_oracle(O, hp, c)
    gi, gr = hp.results()
    assert np.array_equal(gi, ind) and np.array_equal(gr, res)
    want = int(np.argmin(np.linalg.norm(hp.vectors_f64[1:] - hp.vectors_f64[0], axis=1)))
    assert [int(v) for v in O.fr_to_ints(ind)].index(1) == want
    hp.free()


def test_c5_database_merkle_one_column_shard_of_eight(api, O):
    """The database Merkle commitment of the demo (tests/demo/mod.rs:54: chip_merkle(&database)) at full size, 10,000 x 128 at
    2^18 rows, as rank 3 of 8 column shards: root, two leaf traces inside the shard, sampled commitments."""
    from halo2_vectordb_amd.pipeline import MerkleHotPath
    rank, world = 3, 8
    hp = MerkleHotPath(n=10000, dim=128, k=K18, P=48, seed=20260005, tau=0xC5, col_shard=(rank, world)).setup()
    leaf_cells = 64 * 2256 + 2250
    assert hp.n_cells == 10000 * 128 + 10000 * leaf_cells + 1 + 16383 * (2256 + 2250)   # vectors, leaves, the zero cell of the padding, 2^14 - 1 nodes
    assert hp.n_adv_cols == 5881 and hp.my_cols in (735, 736)
    # poison the stream: a column reading a cell this rank did not trace cannot pass by accident
    assert hp.lib.vdb_memset_dev(hp.d_stream.ptr, 0xA5, hp.n_cells * 32) == 0
    com = hp.step().copy()
    assert np.array_equal(hp.results(), O.poseidon_merkle_root(hp.qvec)), "root differs from the oracle's"
    # leaves whose whole trace lies inside this rank's stretch of the stream
    lo, hi = hp.win_adv
    first = -(-(lo - hp.n_in) // leaf_cells)
    last = (hi - hp.n_in) // leaf_cells - 1
    assert 0 <= first < last < 10000
    for leaf in (first, (first + last) // 2):
        cc = O.Ctx(store=True)
        cc.merkle_commitment(hp.qvec[leaf: leaf + 1])
        got = hp.d_stream.download((leaf_cells, 4), offset=(hp.n_in + leaf * leaf_cells) * 32)
        assert np.array_equal(got, cc.advice()[:leaf_cells]), f"trace of leaf {leaf}"
    hp.relayout()
    mine = hp.global_columns()
    pick = [mine[0], mine[len(mine) // 2], mine[-1]]
    cols = hp.download_columns(pick)
    assert np.array_equal(com[[hp.local_index(p) for p in pick]], O.msm_batch(cols, hp.g_lagrange, threads=8))
    hp.free()
    assert hp.lib.vdb_scratch_release() == 0
