"""Two independently composed provers agree, byte for byte.

GPU: halo2_vectordb_amd (witness kernels -> ProverRounds.keygen / prove(seed=S)), everything through the C ABI.
CPU: oracle/prover.py — `create_proof` assembled from the oracle's bricks on the textbook route (witness by the oracle's Context,
numerator on all 4 n points of the extended domain, one division, one inverse transform; Python-integer transcript and SHPLONK).
Both are given the same circuit description, the same SRS and the same blinding scalars (halo2 draws those from the OS, so bit-exact
proof bytes are only defined with the scalars as input: SURVEY 8 f2); then

    verifying key:  every fixed commitment (selectors, sigma columns, constants, table) and the key's digest equal
    proof:          the bytes equal — advice / permuted / product / quotient commitments, every evaluation, both SHPLONK points

for the circuits VERDICT r03 names: the cosine k-means (the satisfiable k-means, examples/kmeans.rs:48-49) resident and streamed,
the Merkle circuit (no lookup columns: constraint degree 3, one column per product, two quotient pieces) and BASELINE C1
(euclidean_distance on two 4-dim vectors, k = 13).  This also holds the GPU's three-coset quotient against the 4 n one.
Parity unpinned against halo2 itself (the reference holds no proof bytes: SURVEY 8c) — what is pinned here is that the protocol the
GPU runs is the protocol as restated a second time, from the description and not from the code."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TAU = 0x1234567890ABCDEF1234567
FIXED = ("sel", "sigma", "cst", "table")


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init(0)
    return a


def _int(O, a):
    return int(O.fr_to_ints(np.asarray(a).reshape(1, 4))[0])


def _compare(O, PV, hp, pr, cs, stream, lookup, seeds, threads=8):
    """keygen and one proof per seed on both sides"""
    g, gl = O.srs_from_tau(hp.k, TAU)
    assert np.array_equal(gl, hp.g_lagrange) and np.array_equal(g, hp.g_monomial)            # the same SRS
    assert (cs.n_adv, cs.n_lk, cs.n_sets, cs.chunk_len, cs.n_h) == (pr.n_adv, pr.n_lk, pr.n_sets, pr.chunk_len, pr.n_h)
    assert cs.instance_cells == pr.instance_cells and cs.consts == pr.consts
    pk = PV.keygen(cs, g, gl, threads=threads)
    for name in FIXED:
        assert np.array_equal(pk.commits[name], pr.fixed[name].commits), name
    assert pk.vk_digest == _int(O, pr.vk_digest())
    outs = []
    for seed in seeds:
        got = pr.prove(None, seed=seed)
        want = PV.prove(pk, stream, lookup, PV.seeded_blinds(cs, seed))
        assert got["instances"] == want["instances"]
        for name in ("adv", "pa", "ps", "zp", "zl", "rand", "h", "hf"):
            assert np.array_equal(got["commitments"][name], want["commitments"][name]), name
        for name in ("beta", "gamma", "y", "x", "yo", "v", "u"):
            assert _int(O, got["challenges"][name]) == want["challenges"][name], name
        assert got["proof"] == want["proof"]
        outs.append(got)
    return pk, outs


def test_cosine_kmeans_resident_and_streamed(api, O):
    from halo2_vectordb_amd import circuit_sym as CS
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import prover as PV
    n, dim, K, I, k, P, L = 8, 4, 2, 1, 12, 48, 11
    cfg = dict(n=n, dim=dim, K=K, I=I, k=k, P=P, L=L, metric="cosine", tau=TAU)
    # the CPU side's circuit and witness: the oracle's Context and the symbolic trace, nothing from the GPU
    hp0 = KmeansHotPath(**cfg)
    vec, _ = hp0._input_vectors()
    qv = O.quantize(vec, P)
    c = O.Ctx(store=True, keygen=True, plan_k=k)
    c.assign_witnesses(qv)
    c.kmeans("cosine", qv, K, I, P=P, L=L)
    stream, lookup = c.advice(), c.lookup()
    # (build_kmeans: the unit blocks placed by the host's numpy builder — the construction the device's is held to entry for entry in
    #  tests/test_gpu_copymap.py; the GPU side below places them with its kernels)
    cm, (cent, _ind) = CS.build_kmeans("cosine", n, dim, K, I, P, L, builder=None)
    public = [int(x) for x in np.asarray(cent).reshape(-1)]                 # examples/kmeans.rs:51-56
    cs = PV.Circuit(k, L, c.break_points(), c.selectors(), len(lookup), cm.copy_of, cm.const_idx, cm.consts, cm.lookup_src, public)
    proofs = {}
    for label, ext_block, block_cols in (("resident", None, 510), ("streamed", 7, 12)):
        hp = KmeansHotPath(**cfg)
        hp.ext_block_cols = ext_block
        hp.setup()
        pr = ProverRounds(hp, block_cols=block_cols).keygen()
        try:
            assert (hp.ext_cols >= hp.n_cols + 2) == (label == "resident")
            assert pr.keygen_report.violations() == 0
            if label == "resident":
                _pk, outs = _compare(O, PV, hp, pr, cs, stream, lookup, seeds=(31, 32))
                assert outs[0]["proof"] != outs[1]["proof"]                # other blinds, another proof
                proofs[label] = outs[0]["proof"]
                pr.fixed_cosets_resident = False                           # the sigma / selector cosets block by block (what C4' does on one card)
                assert pr.prove(None, seed=31)["proof"] == proofs[label]
            else:
                proofs[label] = pr.prove(None, seed=31)["proof"]
        finally:
            pr.free()
            hp.free()
    assert proofs["streamed"] == proofs["resident"]


def test_merkle_circuit_degree_three(api, O):
    from halo2_vectordb_amd.copymap import merkle_circuit_map
    from halo2_vectordb_amd.pipeline import MerkleHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import prover as PV
    n, dim, k = 6, 5, 11
    hp = MerkleHotPath(n=n, dim=dim, k=k, tau=TAU).setup()
    pr = ProverRounds(hp)
    pr.map_on_device = False          # the host's construction of the map (copymap.merkle_circuit_map), the one the CPU side is given below:
    pr.keygen()                       # the device's placement numbers the constants in another order — another, equally valid, fixed column
    try:
        qv = O.quantize(hp.vectors_f64, hp.P)
        c = O.Ctx(store=True, keygen=True, plan_k=k)
        c.assign_witnesses(qv)
        root = c.merkle_commitment(qv)
        stream = c.advice()
        # the circuit's description: gate starts from the oracle; the constant cells' positions are the kernels' flag bytes (the
        # oracle's Context records gate starts only) — data independent, part of the circuit, not of the witness
        d_flags = hp.keygen_flags()
        flags = d_flags.download((hp.n_cells,), dtype=np.uint8)
        d_flags.free()
        assert np.array_equal(flags & 1, c.selectors())
        ints = O.fr_to_ints(stream)
        cm, root_cell = merkle_circuit_map(n, dim, flags, lambda lo, hi: ints[lo:hi])
        assert np.array_equal(stream[root_cell], root)
        cs = PV.Circuit(k, hp.L, c.break_points(), c.selectors(), 0, cm.copy_of, cm.const_idx, cm.consts, None, [int(root_cell)])
        assert (cs.degree, cs.chunk_len, cs.n_h) == (3, 1, 2) and pr.degree == 3
        _pk, outs = _compare(O, PV, hp, pr, cs, stream, np.zeros((0, 4), dtype=np.uint64), seeds=(4,))
        assert outs[0]["instances"] == [_int(O, root)]                     # examples/merkle.rs:47: the root is the statement
    finally:
        pr.free()
        hp.free()


def test_baseline_c1_euclidean_distance_k13(api, O):
    """BASELINE configs[0]: euclidean_distance on two 4-dim vectors, LOOKUP_BITS = 12, k = 13 (data/distances.in's values extended to 4
    dims), the distance public (examples/distances.rs:44-47)"""
    from halo2_vectordb_amd import circuit_sym as CS
    from halo2_vectordb_amd.pipeline import DistancesHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import prover as PV
    a, b = [0.123, 0.456, 1.789, 1.123], [1.123, 0.456, 0.789, 0.123]
    k, P, L = 13, 48, 12
    hp = DistancesHotPath(dim=4, metrics=("euclidean",), k=k, P=P, L=L, tau=TAU, vectors=np.array([a, b])).setup()
    pr = ProverRounds(hp).keygen()
    try:
        assert pr.keygen_report.violations() == 0 and (pr.n_adv, pr.n_lk) == (3, 1)          # 3 advice + 1 lookup column
        qa, qb = O.quantize(a, P), O.quantize(b, P)
        c = O.Ctx(store=True, keygen=True, plan_k=k)
        c.assign_witnesses(qa)
        c.assign_witnesses(qb)
        dist = c.distance("euclidean", qa, qb, P=P, L=L)
        cm, outs_sym = CS.trace_distances(("euclidean",), 4, P, L)
        cs = PV.Circuit(k, L, c.break_points(), c.selectors(), c.n_lookup, cm.copy_of, cm.const_idx, cm.consts, cm.lookup_src, [int(x) for x in outs_sym])
        _pk, outs = _compare(O, PV, hp, pr, cs, c.advice(), c.lookup(), seeds=(13,))
        assert outs[0]["instances"] == [_int(O, dist)]
        # the bytes committed under tests/golden/ (written by the CPU prover alone: tests/test_oracle_prover_cpu.py)
        import hashlib, json, os
        G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cpu_prover_golden.json")))
        assert hashlib.sha256(outs[0]["proof"]).hexdigest() == G["c1_proof_sha256"] and "%064x" % _int(O, pr.vk_digest()) == G["c1_vk_digest"]
        want = float(np.linalg.norm(np.array(a) - np.array(b)))
        assert abs(float(O.dequantize(dist.reshape(1, 4), P)[0]) - want) <= 1e-6 * want       # tests/distances_test.rs's tolerance
    finally:
        pr.free()
        hp.free()


def test_nearest_vector_circuit(api, O):
    """nearest_vector over 6 x 4 vectors (examples/query.rs:32-58 without the Merkle half; tests/vectordb/mod.rs:220-247 assigns the
    query, then the vectors), the result vector public: the same byte-for-byte comparison"""
    from halo2_vectordb_amd import circuit_sym as CS
    from halo2_vectordb_amd.pipeline import NearestHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import prover as PV
    n, dim, k, P, L = 6, 4, 12, 48, 11
    hp = NearestHotPath(n=n, dim=dim, k=k, P=P, L=L, tau=TAU).setup()
    pr = ProverRounds(hp).keygen()
    try:
        assert pr.keygen_report.violations() == 0
        rows = O.quantize(hp.vectors_f64, P)                    # row 0: the query, rows 1..n: the database
        c = O.Ctx(store=True, keygen=True, plan_k=k)
        c.assign_witnesses(rows[0])
        c.assign_witnesses(rows[1:])
        ind, res = c.nearest_vector("euclidean", rows[0], rows[1:], P=P, L=L)
        cm, (_ind, res_cells) = CS.build_nearest("euclidean", n, dim, P, L, builder=None)
        cs = PV.Circuit(k, L, c.break_points(), c.selectors(), c.n_lookup, cm.copy_of, cm.const_idx, cm.consts, cm.lookup_src, [int(x) for x in res_cells])
        _pk, outs = _compare(O, PV, hp, pr, cs, c.advice(), c.lookup(), seeds=(21,))
        assert outs[0]["instances"] == [int(v) for v in O.fr_to_ints(res)]      # examples/query.rs:58 make_public.extend(result)
        want = int(np.argmin(np.linalg.norm(hp.vectors_f64[1:] - hp.vectors_f64[0], axis=1)))
        assert [int(v) for v in O.fr_to_ints(ind)].index(1) == want
    finally:
        pr.free()
        hp.free()


def test_query_circuit_nearest_and_merkle_in_one(api, O):
    """examples/query.rs:32-73 (tests/vectordb/mod.rs:220-247 chip_nearest_vector): nearest_vector(query, database) and
    merkle_commitment(database) over the same assigned vectors in ONE circuit, the result vector and then the root public.  The map of
    this circuit is placed block by block on the device (circuit_dev.py; tests/test_gpu_copymap.py holds that placement against the
    host's construction); the CPU prover is handed its arrays as downloaded and the oracle's own witness, break points and gate starts."""
    from halo2_vectordb_amd.pipeline import QueryHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import prover as PV
    n, dim, k, P, L = 5, 4, 12, 48, 11
    hp = QueryHotPath(n=n, dim=dim, k=k, P=P, L=L, tau=TAU).setup()
    pr = ProverRounds(hp).keygen()
    try:
        assert pr.keygen_report.violations() == 0
        rows = O.quantize(hp.vectors_f64, P)                    # row 0: the query, rows 1..n: the database
        c = O.Ctx(store=True, keygen=True, plan_k=k)
        c.assign_witnesses(rows[0])
        c.assign_witnesses(rows[1:])
        ind, res = c.nearest_vector("euclidean", rows[0], rows[1:], P=P, L=L)
        root = c.merkle_commitment(rows[1:])
        stream = c.advice()
        assert stream.shape[0] == hp.n_cells and c.n_lookup == hp.n_lookup
        cm = pr.circuit
        cs = PV.Circuit(k, L, c.break_points(), c.selectors(), c.n_lookup, cm.copy_of, cm.const_idx, cm.consts, cm.lookup_src, list(pr.instance_cells))
        assert len(pr.instance_cells) == dim + 1 and np.array_equal(stream[pr.instance_cells[-1]], root)
        _pk, outs = _compare(O, PV, hp, pr, cs, stream, c.lookup(), seeds=(34,))
        assert outs[0]["instances"] == [int(v) for v in O.fr_to_ints(res)] + [_int(O, root)]     # examples/query.rs:58, :69
        want = int(np.argmin(np.linalg.norm(hp.vectors_f64[1:] - hp.vectors_f64[0], axis=1)))
        assert [int(v) for v in O.fr_to_ints(ind)].index(1) == want
    finally:
        pr.free()
        hp.free()


def test_random_kmeans_circuits_prove_the_same_bytes_on_both_provers(api, O):
    """The comparison above at shapes nobody picked: seeded random k-means circuits (vectors, clusters, iterations, distance, lookup
    width, column height, block size of the streamed rounds) — the numbers of advice and lookup columns, whether the last set of the
    permutation argument is full, where the set that spans the advice / lookup junction falls all vary.  Satisfiable distances only
    (cosine, manhattan, hamming: the Euclidean k-means violates qlog2's asserted constants at iteration 0, tests/test_gpu_copymap.py)."""
    from halo2_vectordb_amd import circuit_sym as CS
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import prover as PV
    rng = np.random.default_rng(31337)
    shapes = set()
    for case in range(4):
        metric = ("cosine", "manhattan", "hamming", "cosine")[case]
        n, dim = int(rng.integers(4, 9)), int(rng.integers(2, 6))
        K, I = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        k, L = (11, int(rng.integers(8, 11))) if case % 2 else (12, int(rng.integers(9, 12)))
        P = 48
        if metric == "hamming":
            vec = np.repeat(np.array([[0.5] * dim, [1.25] * dim]), (n + 1) // 2, axis=0)[:n]
            vec = vec[[i // 2 + (i % 2) * ((n + 1) // 2) for i in range(n)]]          # two groups, alternating: equal elements exist
            for i in range(2, n):
                vec[i, int(rng.integers(0, dim))] += 0.125 * i
        else:
            vec = rng.uniform(0.5, 3.0, size=(n, dim))
        cfg = dict(n=n, dim=dim, K=K, I=I, k=k, P=P, L=L, metric=metric, tau=TAU, vectors=vec)
        qv = O.quantize(vec, P)
        c = O.Ctx(store=True, keygen=True, plan_k=k)
        c.assign_witnesses(qv)
        c.kmeans(metric, qv, K, I, P=P, L=L)
        assert c.err == 0, (case, metric)
        cm, (cent, _ind) = CS.build_kmeans(metric, n, dim, K, I, P, L, builder=None)
        cs = PV.Circuit(k, L, c.break_points(), c.selectors(), c.n_lookup, cm.copy_of, cm.const_idx, cm.consts, cm.lookup_src, [int(x) for x in np.asarray(cent).reshape(-1)])
        hp = KmeansHotPath(**cfg)
        if case % 2:
            hp.ext_block_cols = int(rng.integers(3, 9))          # the streamed rounds, with a block size of its own
        hp.setup()
        pr = ProverRounds(hp, block_cols=int(rng.integers(4, 20)) * 2 if case % 2 else 510).keygen()
        try:
            assert pr.keygen_report.violations() == 0, (case, metric, pr.keygen_report.as_dict())
            _compare(O, PV, hp, pr, cs, c.advice(), c.lookup(), seeds=(100 + case,), threads=8)
            shapes.add((pr.n_adv, pr.n_lk, pr.n_sets))
        finally:
            pr.free()
            hp.free()
    assert len(shapes) >= 3
