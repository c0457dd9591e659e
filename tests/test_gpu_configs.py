"""BASELINE.json configs at (or near) their full sizes, through the C ABI.

C1 euclidean_distance on two 4-dim vectors, k=13 (LOOKUP_BITS=12): full stream + MockProver-like gate check.
C2 nearest_vector over 64 x 128-dim SIFT-shaped vectors (LOOKUP_BITS=13): full stream vs the oracle.
C3 merkle_commitment over 1024 x 128 (154 M trace cells): size-independent properties — the trace kernel's root
   equals the hash-only kernel's root (two independent GPU code paths) and the oracle's value-level root; sampled
   leaf traces equal the oracle's trace of that leaf.
C4 k=16 MSM: exact closed form with known discrete logs (65,536 points).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init()
    return a


def sift(seed, n, dim):
    v = np.random.default_rng(seed).integers(0, 219, size=(n, dim)).astype(np.float64)
    v[v.sum(axis=1) == 0, 0] = 1
    return v


def test_c1_euclid_k13_mock(api, O):
    a, b = [0.123, 0.456, 1.789, 1.123], [1.123, 0.456, 0.789, 0.123]   # data/distances.in extended to 4 dims
    qa, qb = api.quantize([a]), api.quantize([b])
    got = api.wit_distance("euclidean", qa, qb, L=12, selectors=True)
    c = O.Ctx(store=True, keygen=True, plan_k=13)
    c.assign_witnesses(qa[0])
    c.assign_witnesses(qb[0])
    r = c.distance("euclidean", qa[0], qb[0], L=12)
    assert np.array_equal(got["result"][0], r)
    assert np.array_equal(got["stream"], c.advice()[8:]) and np.array_equal(got["lookup"], c.lookup())
    assert c.check_gates(12) == 0                      # every gate row and lookup cell satisfied
    want = float(np.linalg.norm(np.array(a) - np.array(b)))
    assert abs(float(api.dequantize(got["result"])[0]) - want) <= 1e-6 * want
    # 3 advice + 1 lookup column at k = 13
    assert len(c.break_points()) + 1 == 3 and -(-c.n_lookup // (8192 - 9)) == 1


def test_c2_nearest_64x128(api, O):
    db, q = sift(20260002, 64, 128), sift(20261002, 1, 128)[0]
    qdb, qq = api.quantize(db), api.quantize(q)
    got = api.wit_nearest("euclidean", qq, qdb, L=13)
    c = O.Ctx(store=True)
    ind, res = c.nearest_vector("euclidean", qq, qdb, L=13)
    assert np.array_equal(got["indicator"], ind) and np.array_equal(got["result"], res)
    assert np.array_equal(got["stream"], c.advice()) and np.array_equal(got["lookup"], c.lookup())
    want = int(np.argmin(np.linalg.norm(db - q, axis=1)))
    assert [int(v) for v in O.fr_to_ints(ind)].index(1) == want


def test_c3_merkle_1024x128(api, O):
    v = api.quantize(sift(20260003, 1024, 128), 32)   # examples/merkle.rs uses PRECISION_BITS = 32
    got = api.wit_merkle(v)
    assert got["stream"].shape[0] == 1024 * (64 * 2256 + 2250) + 1023 * (2256 + 2250)
    assert np.array_equal(got["root"], api.poseidon_merkle_root(v))
    assert np.array_equal(got["root"], O.poseidon_merkle_root(v))
    leaf_cells = 64 * 2256 + 2250
    for leaf in (0, 517, 1023):
        c = O.Ctx(store=True)
        c.merkle_commitment(v[leaf:leaf + 1])
        assert np.array_equal(got["stream"][leaf * leaf_cells:(leaf + 1) * leaf_cells], c.advice()[:leaf_cells])


def test_c4_msm_k16_closed_form(api, O):
    k, n = 16, 1 << 16
    rng = np.random.default_rng(20260004)
    hs = [int(x) for x in rng.integers(1, 1 << 62, size=n)]
    bases = O.g1_mul_generator(hs)
    srs = api.Srs(k, None, bases)
    vals = rng.integers(0, 1 << 15, size=n)
    vals[rng.random(n) < 0.4] = 0
    vals[rng.random(n) < 0.1] = 1
    small = [int(x) for x in vals]
    neg = [(R - int(x)) % R for x in rng.integers(0, 1 << 60, size=n)]
    cols = np.stack([O.fr_from_ints(small), O.fr_from_ints(neg), O.random_fr(rng, n)])
    got = api.msm_batch(srs, cols, basis=1)
    ks = [sum(x * h for x, h in zip(O.fr_to_ints(cc), hs)) % R for cc in cols]
    assert np.array_equal(got, O.g1_mul_generator(ks))
    srs.free()


def test_constant_factoring_is_data_independent(api, O):
    """Keygen with database A, prove database B: the commitments computed with the constant cells factored out
    (vdb_msm_batch_masked_dev + keygen-time constant points) must equal the plain MSM of B's real columns."""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    hp = KmeansHotPath(n=14, dim=6, K=3, I=2, k=10, P=48, L=9, seed=11, blind_seed=3).setup()   # fixed blinds: commitments are compared across steps
    assert hp.const_cell_fraction is not None and 0.15 < hp.const_cell_fraction < 0.6
    rng = np.random.default_rng(999)
    hp.set_vectors(rng.integers(0, 219, size=(14, 6)).astype(np.float64) + rng.random((14, 6)))
    got = hp.step().copy()
    hp.relayout()
    cols = hp.download_columns(list(range(hp.n_cols)))
    want = O.msm_batch(cols, hp.g_lagrange, threads=4)
    assert np.array_equal(got, want)
    hp.factor_constants = False
    assert np.array_equal(hp.step(), want)
    # the stream itself matches the oracle for database B
    qv = O.quantize(hp.vectors_f64, 48)
    c = O.Ctx(store=True)
    c.assign_witnesses(qv)
    c.kmeans("euclidean", qv, 3, 2, P=48, L=9)
    assert np.array_equal(hp.d_stream.download((hp.n_cells, 4)), c.advice())
    hp.free()


def test_constant_flags_mark_only_constants(api, O):
    """cells flagged constant must hold the same value for two different inputs of the same shape"""
    rng = np.random.default_rng(5)
    for metric in ("euclidean", "cosine", "manhattan"):
        a1, b1 = O.quantize(rng.uniform(-3, 3, (2, 7))), O.quantize(rng.uniform(-3, 3, (2, 7)))
        a2, b2 = O.quantize(rng.uniform(-300, 300, (2, 7))), O.quantize(rng.uniform(-300, 300, (2, 7)))
        g1 = api.wit_distance(metric, a1, b1, selectors=True)
        g2 = api.wit_distance(metric, a2, b2, selectors=True)
        assert np.array_equal(g1["flags"], g2["flags"])
        m = g1["const_mask"].astype(bool)
        assert m.mean() > 0.15
        assert np.array_equal(g1["stream"][m], g2["stream"][m])
    v1, v2 = O.quantize(rng.random((5, 4)), 32), O.quantize(rng.random((5, 4)) * 7, 32)
    m1, m2 = api.wit_merkle(v1, selectors=True), api.wit_merkle(v2, selectors=True)
    m = m1["const_mask"].astype(bool)
    assert np.array_equal(m1["flags"], m2["flags"]) and m.mean() > 0.15
    assert np.array_equal(m1["stream"][m], m2["stream"][m])


@pytest.mark.parametrize("world", [2, 3, 5])
@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_sharded_ranks_reproduce_the_unsharded_job(api, O, world, metric):
    """every rank commits its block of advice / lookup columns from a witness it generated only for those
    columns (vdb_wit_set_window).  Emulated here rank by rank on one GPU: the columns and commitments of all ranks,
    reassembled in the order gather_commitments uses, are exactly those of the unsharded job, and the k-means results
    (computed on the value-only path everywhere) are the same on every rank."""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    cfg = dict(n=20, dim=9, K=3, I=3, k=10, P=48, L=9, seed=17, metric=metric, blind_seed=4)   # the ranks must blind alike to be compared
    full = KmeansHotPath(**cfg).setup()
    want_commit = full.step().copy()
    full.relayout()
    want_cols = full.download_columns(list(range(full.n_cols)))
    want_cent, want_ind = full.results()
    assert full.n_adv_cols >= world and full.n_lk_cols >= 1
    full.free()
    adv, lk = [], []
    for r in range(world):
        hp = KmeansHotPath(col_shard=(r, world), **cfg).setup()
        # poison the stream so that a column reading a cell its rank did not emit cannot pass by accident
        assert hp.lib.vdb_memset_dev(hp.d_stream.ptr, 0xA5, hp.n_cells * 32) == 0
        assert hp.lib.vdb_memset_dev(hp.d_lookup.ptr, 0xA5, max(hp.n_lookup, 1) * 32) == 0
        got = hp.step().copy()
        adv.append(got[: hp.my_adv])
        lk.append(got[hp.my_adv:])
        hp.relayout()
        mine = hp.global_columns()
        assert np.array_equal(hp.download_columns(mine), want_cols[mine]), f"rank {r} columns"
        cent, ind = hp.results()
        assert np.array_equal(cent, want_cent) and np.array_equal(ind, want_ind)
        hp.free()
    assert np.array_equal(np.concatenate(adv + lk), want_commit)


def test_sharded_kmeans_with_the_hamming_distance(api, O):
    """the same for the distance that counts equal elements (distance.rs:146-175; its value-only form serves the ranks that store
    other columns): two groups of vectors that agree on most coordinates, so that equal elements exist at all — against the oracle's
    cells on one rank, and two ranks against one"""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    rng = np.random.default_rng(8)
    vec = np.repeat(np.array([[0.25] * 6, [1.5] * 6]), 6, axis=0)[[0, 6, 1, 7, 2, 8, 3, 9, 4, 10, 5, 11]]      # A, B, A, B, ...
    for i in range(2, 12):
        vec[i, rng.integers(0, 6)] += 0.125 * (1 + i)                      # one coordinate of its own each
    cfg = dict(n=12, dim=6, K=2, I=2, k=10, P=48, L=9, metric="hamming", blind_seed=4, vectors=vec)
    full = KmeansHotPath(**cfg).setup()
    want_commit = full.step().copy()
    want_cent, want_ind = full.results()
    c = O.Ctx(store=True)
    q = O.quantize(vec, 48)
    c.assign_witnesses(q)
    cent, ind = c.kmeans("hamming", q, 2, 2, P=48, L=9)
    assert np.array_equal(want_cent, cent) and np.array_equal(want_ind, ind)
    assert np.array_equal(full.d_stream.download((full.n_cells, 4)), c.advice()) and np.array_equal(full.d_lookup.download((full.n_lookup, 4)), c.lookup())
    assert sorted(np.array(O.fr_to_ints(ind.reshape(-1, 4)), dtype=object).reshape(12, 2).sum(axis=0).tolist()) == [6 << 48, 6 << 48]   # both clusters keep their six
    full.relayout()
    want_cols = full.download_columns(list(range(full.n_cols)))
    full.free()
    got = []
    for r in range(2):
        hp = KmeansHotPath(col_shard=(r, 2), **cfg).setup()
        assert hp.lib.vdb_memset_dev(hp.d_stream.ptr, 0xA5, hp.n_cells * 32) == 0
        assert hp.lib.vdb_memset_dev(hp.d_lookup.ptr, 0xA5, max(hp.n_lookup, 1) * 32) == 0
        com = hp.step().copy()
        got.append((com[: hp.my_adv], com[hp.my_adv:]))
        hp.relayout()
        mine = hp.global_columns()
        assert np.array_equal(hp.download_columns(mine), want_cols[mine]), f"rank {r} columns"
        cent_r, ind_r = hp.results()
        assert np.array_equal(cent_r, want_cent) and np.array_equal(ind_r, want_ind)
        hp.free()
    assert np.array_equal(np.concatenate([g[0] for g in got] + [g[1] for g in got]), want_commit)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_nearest_vector_ranks_reproduce_the_unsharded_job(api, O, world):
    """SURVEY §8(e) for nearest_vector: "gather N distances then the short min chain" — here every rank computes the N distance
    values and the minimum chain itself (value-only walk, no exchange) and stores only the cells of its own columns.  Ranks
    emulated one after another on one GPU with a poisoned stream: columns, commitments, indicator and result vector equal the
    unsharded job's."""
    from halo2_vectordb_amd.pipeline import NearestHotPath
    cfg = dict(n=9, dim=6, k=11, P=48, L=10, seed=23, blind_seed=4)
    full = NearestHotPath(**cfg).setup()
    want_commit = full.step().copy()
    full.relayout()
    want_cols = full.download_columns(list(range(full.n_cols)))
    want_ind, want_res = full.results()
    assert full.n_adv_cols >= 2 * world and full.n_lk_cols >= 1
    full.free()
    adv, lk, stored = [], [], []
    for r in range(world):
        hp = NearestHotPath(col_shard=(r, world), **cfg).setup()
        assert hp.shard_witness
        assert hp.lib.vdb_memset_dev(hp.d_stream.ptr, 0xA5, hp.n_cells * 32) == 0
        assert hp.lib.vdb_memset_dev(hp.d_lookup.ptr, 0xA5, max(hp.n_lookup, 1) * 32) == 0
        got = hp.step().copy()
        adv.append(got[: hp.my_adv])
        lk.append(got[hp.my_adv:])
        hp._witness()
        api.sync()
        poison = np.full(4, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)
        stored.append(int((hp.d_stream.download((hp.n_cells, 4)) != poison).any(axis=1).sum()))
        hp.relayout()
        mine = hp.global_columns()
        assert np.array_equal(hp.download_columns(mine), want_cols[mine]), f"rank {r} columns"
        ind, res = hp.results()
        assert np.array_equal(ind, want_ind) and np.array_equal(res, want_res)
        hp.free()
    assert np.array_equal(np.concatenate(adv + lk), want_commit)
    assert max(stored) < 0.75 * sum(stored)               # no rank stores (nearly) the whole witness any more


def test_pinning_file_keygen_then_prove(api, O, tmp_path):
    """Keygen arm writes configs/{name}.json, Prove arm reads it back (src/scaffold/mod.rs:272, 285-287): same break
    points, same commitments; a pinning of another circuit is refused."""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    cfg = dict(n=14, dim=6, K=3, I=2, k=10, P=48, L=9, seed=11, blind_seed=5)
    keygen = KmeansHotPath(**cfg).setup()
    pin = tmp_path / "kmeans.json"
    keygen.write_pinning(pin)
    want = keygen.step().copy()
    keygen.free()
    prove = KmeansHotPath(**cfg).setup(pinning=pin)
    assert np.array_equal(prove.step(), want)
    prove.free()
    other = KmeansHotPath(**dict(cfg, I=1))
    with pytest.raises(ValueError):
        other.setup(pinning=pin)
    other.free()


@pytest.mark.parametrize("k,metric", [(11, "euclidean"), (13, "euclidean"), (12, "cosine"), (11, "manhattan")])
def test_virtual_layout_matches_the_copied_layout(api, O, k, metric):
    """committing and transforming straight from the witness stream (vdb_colsrc) gives the same commitments, coefficient
    columns and extended columns as the path that first copies the stream into columns"""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    cfg = dict(n=14, dim=6, K=3, I=2, k=k, P=48, L=9, seed=11, metric=metric, blind_seed=6)
    out = {}
    for virt in (False, True):
        hp = KmeansHotPath(**cfg)
        hp.virtual_layout = virt
        hp.setup()
        com = hp.step().copy()
        coeff = hp.d_cols.download((hp.my_cols, hp.rows, 4))
        ext = hp.d_ext.download((hp.my_cols, 4 * hp.rows, 4))
        out[virt] = (com, coeff, ext)
        if virt:   # and against the oracle, from the columns the copy path lays out
            hp.relayout()
            cols = hp.download_columns(list(range(hp.n_cols)))
            assert np.array_equal(com, O.msm_batch(cols, hp.g_lagrange, threads=4))
            wc, we = O.lde_batch(cols[:3], ext=2, threads=3)
            assert np.array_equal(coeff[:3], wc) and np.array_equal(ext[:3], we)
        hp.free()
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a, b)


def test_merkle_hot_path_c3_and_column_shards(api, O):
    """C3 (merkle_commitment over 1024 x 128 at k = 15) through the resident hot path: the root the trace ends in is the
    hash-only root, sampled commitments equal the oracle's MSM of the laid-out columns, and two column shards reproduce
    the unsharded commitments (the way C5's 2^18-row circuits are cut to fit: tools/c5_subcircuits.py)."""
    from halo2_vectordb_amd.pipeline import MerkleHotPath
    hp = MerkleHotPath(n=1024, dim=128, k=15, tau=0x5EED, blind_seed=7).setup()
    assert hp.n_lk_cols == 0 and hp.n_cells == 1024 * 128 + 1024 * (64 * 2256 + 2250) + 1023 * (2256 + 2250)    # assigned vectors + trace
    full = hp.step().copy()
    assert np.array_equal(hp.results(), api.poseidon_merkle_root(hp.qvec))
    hp.relayout()
    pick = [0, hp.n_adv_cols // 2, hp.n_adv_cols - 1]
    assert np.array_equal(full[pick], O.msm_batch(hp.download_columns(pick), hp.g_lagrange))
    n_adv = hp.n_adv_cols
    hp.free()
    parts = []
    for rank in range(2):
        h = MerkleHotPath(n=1024, dim=128, k=15, tau=0x5EED, col_shard=(rank, 2), blind_seed=7).setup()
        parts.append(h.step().copy())
        h.free()
    assert sum(len(p) for p in parts) == n_adv and np.array_equal(np.concatenate(parts), full)


def test_nearest_hot_path_c2(api, O):
    """C2 (nearest_vector over 64 x 128 + query at k = 14) through the resident hot path (vdb_wit_nearest_dev): same cells as the
    host-buffer entry point behind [query | vectors], same indicator / result, sampled commitments equal the oracle's MSM"""
    from halo2_vectordb_amd.pipeline import NearestHotPath
    hp = NearestHotPath(n=64, dim=128, k=14, L=13, tau=0xC2).setup()
    got = hp.step().copy()
    ref = api.wit_nearest("euclidean", hp.qvec[0], hp.qvec[1:], L=13)
    assert hp.n_cells == 65 * 128 + ref["stream"].shape[0] and hp.n_lookup == ref["lookup"].shape[0]
    ind, res = hp.results()
    assert np.array_equal(ind, ref["indicator"]) and np.array_equal(res, ref["result"])
    want = int(np.argmin(np.linalg.norm(hp.vectors_f64[1:] - hp.vectors_f64[0], axis=1)))
    assert [int(v) for v in O.fr_to_ints(ind)].index(1) == want
    hp.relayout()
    pick = [0, 1, hp.n_adv_cols - 1, hp.n_adv_cols, hp.n_cols - 1]
    cols = hp.download_columns(pick)
    assert np.array_equal(cols[0][: 65 * 128], hp.qvec.reshape(-1, 4)[: 65 * 128])
    assert np.array_equal(got[pick], O.msm_batch(cols, hp.g_lagrange))
    hp.free()


def test_two_rank_timed_path_with_the_collective(tmp_path):
    """The N > 1 path of bench.py exactly as the driver launches it (torch.distributed.run, one process per rank), here
    with two ranks sharing this box's one GPU over gloo (VDB_DIST_BACKEND=gloo; on a multi-GPU node the same code runs over
    RCCL): every timed step ends with the all_gather of the commitments, and the gathered commitments equal the unsharded
    job's (--verify-gather)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VDB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--small", "--no-cpu-baseline", "--verify-gather"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["gathered_commitments_match_unsharded_job"] is True
    assert out["value"] > 0 and out["roofline"]["launches_per_step"] > 0
    # ... and the whole proof that follows the timed region ran sharded over the same two ranks (rounds.ProverRounds with a dist.Comm)
    pf = out["proof"]
    assert pf["n_gpus"] == 2 and pf["every_rank_wrote_the_same_proof_bytes"] and pf["quotient_identity_at_x_holds"] and pf["mock_prover_violations"] == 0
    assert pf["proof_ms"] > 0 and pf["n_instances"] == 2 * 8


def test_a_failing_rank_of_the_sharded_proof_does_not_cost_the_bench_line(tmp_path):
    """bench.py with N > 1: the sharded proof runs in a worker thread with a bounded wait.  A rank that drops out of it (injected
    here) leaves the other rank waiting in a collective; the timed hot-path line is printed all the same, with the failure in
    `proof`, and every rank exits cleanly."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VDB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", VDB_BENCH_TEST_FAIL_RANK="1", VDB_BENCH_PROOF_TIMEOUT="5")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29519",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--small", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "error" in out["proof"] and out["proof_abandoned"] is True
    assert "the sharded proof was abandoned" in res.stderr
    # ... and with VDB_BENCH_STRICT_EXIT=1 the launcher sees the failure as a non-zero exit code, the line still printed first
    cmd[cmd.index("29519")] = "29520"
    res = subprocess.run(cmd, env=dict(env, VDB_BENCH_STRICT_EXIT="1"), capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["value"] > 0 and out["proof_abandoned"] is True


def test_extended_cosets_in_column_blocks(api, O):
    """A circuit whose cosets do not fit HBM streams through: coeff_to_extended runs block after block into one buffer.
    Forced here on a small circuit: same commitments and coefficients as the resident run, and the buffer ends up holding
    the cosets of the last block, equal to the resident run's cosets of those columns and to the oracle's."""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    cfg = dict(n=14, dim=6, K=3, I=2, k=11, P=48, L=9, seed=11, blind_seed=8)
    full = KmeansHotPath(**cfg).setup()
    assert full.ext_cols == full.n_cols + 2                      # everything fits: resident
    com = full.step().copy()
    ext_all = full.d_ext.download((full.n_cols, 4 * full.rows, 4))
    coeff = full.d_cols.download((full.n_cols, full.rows, 4))
    full.free()
    hp = KmeansHotPath(**cfg)
    hp.ext_block_cols = 7
    hp.setup()
    assert hp.ext_cols == 7 and hp.n_cols > 3 * 7
    assert np.array_equal(hp.step(), com) and np.array_equal(hp.d_cols.download((hp.n_cols, hp.rows, 4)), coeff)
    last0 = (hp.n_cols - 1) // 7 * 7
    nb = hp.n_cols - last0
    got = hp.d_ext.download((nb, 4 * hp.rows, 4))
    assert np.array_equal(got, ext_all[last0:])
    hp.relayout()
    cols = hp.download_columns([last0])
    assert np.array_equal(got[0], O.lde_batch(cols, ext=2)[1][0])
    hp.free()
