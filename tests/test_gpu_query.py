"""The reference's `query` circuit (examples/query.rs:32-73; tests/vectordb/mod.rs:220-247 chip_nearest_vector): nearest_vector and
merkle_commitment over the same assigned database in ONE circuit, the result vector and the Merkle root public — "verifiable vector
similarity queries over a committed vector database", the README's subject — through the hot path and the whole proof."""
import numpy as np
import pytest

from test_gpu_rounds import FIXED, TAU, _meta, _verify

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init(0)
    return a


@pytest.mark.parametrize("metric,n,dim", [("euclidean", 6, 4), ("cosine", 5, 3)])
def test_query_stream_is_the_oracles(api, O, metric, n, dim):
    """[query | vectors | nearest_vector | merkle_commitment]: every advice cell, lookup cell and gate flag equal to the oracle's
    context run the same way; the results are the oracle's (indicator, vector, root); f64 agrees on the index"""
    from halo2_vectordb_amd.pipeline import QueryHotPath
    hp = QueryHotPath(n=n, dim=dim, k=12, L=11, metric=metric, tau=TAU).setup()
    try:
        d_flags = hp.keygen_flags()
        flags = d_flags.download((hp.n_cells,), dtype=np.uint8)
        d_flags.free()
        hp._witness()
        api.sync()
        qv = hp.qvec
        c = O.Ctx(store=True, keygen=True, plan_k=12)
        c.assign_witnesses(qv[0])
        c.assign_witnesses(qv[1:])
        ind, res = c.nearest_vector(metric, qv[0], qv[1:], P=48, L=11)
        root = c.merkle_commitment(qv[1:])
        assert len(c) == hp.n_cells and c.n_lookup == hp.n_lookup
        assert np.array_equal(hp.d_stream.download((hp.n_cells, 4)), c.advice())
        assert np.array_equal(hp.d_lookup.download((hp.n_lookup, 4)), c.lookup())
        assert np.array_equal(flags & 1, c.selectors().astype(np.uint8) & 1)
        g_ind, g_res, g_root = hp.results()
        assert np.array_equal(g_ind, ind) and np.array_equal(g_res, res) and np.array_equal(g_root, root)
        assert np.array_equal(g_root, api.poseidon_merkle_root(qv[1:]))
        v = hp.vectors_f64
        d = np.linalg.norm(v[1:] - v[0], axis=1) if metric == "euclidean" else 1 - (v[1:] @ v[0]) / (np.linalg.norm(v[1:], axis=1) * np.linalg.norm(v[0]))
        assert [int(x) for x in O.fr_to_ints(g_ind)].index(1) == int(np.argmin(d))
    finally:
        hp.free()


def test_query_proof_states_the_result_and_the_root(api, O):
    """the whole proof: every constraint of both gadgets in the permutation argument (the leaves absorb copies of the very cells the
    distances read), result vector and root tied to the instance column in make_public order (examples/query.rs:58, :69); the
    verifier accepts them and rejects another root or another result word"""
    from halo2_vectordb_amd.pipeline import QueryHotPath
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    from oracle import pairing as PR
    hp = QueryHotPath(n=6, dim=4, k=12, L=11, metric="cosine", tau=TAU).setup()
    pr = ProverRounds(hp).keygen()
    try:
        assert pr.keygen_report.violations() == 0, pr.keygen_report.as_dict()
        cm = pr.circuit
        # the leaves' absorbed words are copies of the assigned vector cells (which the distances copy too)
        words = np.arange(hp.dim, hp.n_in)
        assert all(int((cm.copy_of == w).sum()) >= 3 for w in words[:8])
        out = pr.prove(None, seed=17)
        _ind, res, root = hp.results()
        want = O.fr_to_ints(res) + O.fr_to_ints(root.reshape(1, 4))
        assert out["instances"] == want and len(want) == hp.dim + 1
        assert quotient_identity_holds(pr, out["challenges"], out["evals"], out["instances"])
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU))
        assert _verify(O, api, out["proof"], {**vk, "instances": want})
        for i in (0, hp.dim):                                      # a word of the result, the root
            other = list(want)
            other[i] = (other[i] + 1) % O.R_MOD
            assert not _verify(O, api, out["proof"], {**vk, "instances": other})
        # another database under the same key
        rng = np.random.default_rng(5)
        hp.set_vectors(rng.integers(0, 219, size=(7, 4)).astype(np.float64) + rng.random((7, 4)))
        assert pr.mock_check().violations() == 0
        out2 = pr.prove(None)
        assert out2["instances"] != want and _verify(O, api, out2["proof"], {**vk, "opened": out2["opened"], "instances": out2["instances"]})
    finally:
        pr.free()
        hp.free()


@pytest.mark.parametrize("n,dim", [(6, 5), (5, 4), (8, 3)])
def test_merkle_map_built_on_the_device_is_the_host_built_map(api, O, n, dim):
    """circuit_dev.place_merkle (the Poseidon trace's copy constraints placed by vdb_copymap_place_dev) against
    copymap.merkle_circuit_map (numpy): same copies, constants, gate flags, root cell — odd and even word counts, padded trees"""
    from halo2_vectordb_amd.pipeline import MerkleHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    hp = MerkleHotPath(n=n, dim=dim, k=11, tau=TAU).setup()
    maps = {}
    try:
        for on_dev in (True, False):
            pr = ProverRounds(hp)
            pr.map_on_device = on_dev
            d_flags = hp.keygen_flags()
            cm = pr.circuit_map(d_flags)
            d_flags.free()
            maps[on_dev] = (cm.copy_of.copy(), cm.const_idx.copy(), np.asarray(cm.gate).copy(), [int(v) for v in cm.consts], pr.root_cell, pr.public_cells)
            if on_dev:
                cm.free()
        d, h = maps[True], maps[False]
        assert d[4] == h[4] and d[5] == h[5]
        assert np.array_equal(d[0], h[0]) and np.array_equal(d[2], h[2])
        # the constants may be numbered differently: compare the values the cells are tied to
        val = lambda m: np.where(m[1] >= 0, np.asarray(m[3] + [0], dtype=object)[np.maximum(m[1], -1)], -1)
        assert np.array_equal(val(d), val(h))
    finally:
        hp.free()
