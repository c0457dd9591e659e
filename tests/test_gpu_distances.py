"""The reference's two-vector circuits (examples/distances.rs:29-59, examples/euclid.rs:26-46; tests/distances/mod.rs chip_*; BASELINE
configs[0]) through the hot path and the whole proof: cells against the oracle, results against f64 at the reference tests' tolerance,
the distances public as `make_public.push(dist)` makes them."""
import numpy as np
import pytest

import examples_common as E
from test_gpu_rounds import FIXED, TAU, _meta, _verify

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init(0)
    return a


def _same_stream(hp, c, O):
    d_flags = hp.keygen_flags()
    flags = d_flags.download((hp.n_cells,), dtype=np.uint8)
    d_flags.free()
    hp._witness()
    assert np.array_equal(hp.d_stream.download((hp.n_cells, 4)), c.advice()[: hp.n_cells])
    assert np.array_equal(hp.d_lookup.download((hp.n_lookup, 4)), c.lookup()[: hp.n_lookup])
    assert np.array_equal(flags & 1, c.selectors().astype(np.uint8)[: hp.n_cells] & 1)


def test_distances_example_proves_with_its_four_public_distances(api, O):
    """examples/distances.rs on data/distances.in (k and LOOKUP_BITS of the README's command line): [a | b | euclidean | manhattan |
    cosine | hamming] is the oracle's whole context; the four distances are the proof's instances"""
    from halo2_vectordb_amd.pipeline import DistancesHotPath
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    from oracle import pairing as PR
    d, cfg = E.load("distances"), E.README["distances"]
    r = E.oracle_distances(O)
    hp = DistancesHotPath(dim=len(d["a"]), k=cfg["k"], L=cfg["L"], tau=TAU, vectors=np.array([d["a"], d["b"]], dtype=np.float64)).setup()
    pr = None
    try:
        assert hp.n_cells == len(r["ctx"]) and hp.n_lookup == r["ctx"].n_lookup and hp.n_cells > r["offsets"][("hamming", 3)][0]
        _same_stream(hp, r["ctx"], O)
        want = [r["results"][(m, i)] for i, m in enumerate(("euclidean", "manhattan", "cosine", "hamming"))]
        assert np.array_equal(hp.results(), np.stack(want))
        for m, got in zip(hp.metrics, api.dequantize(hp.results())):
            f = E.F64[m](d["a"], d["b"])
            assert abs(float(got) - f) <= 1e-6 * max(abs(f), 1.0)              # assert_float_relative_eq! of tests/distances_test.rs
        pr = ProverRounds(hp).keygen()
        assert pr.keygen_report.violations() == 0, pr.keygen_report.as_dict()
        assert len(pr.instance_cells) == 4
        out = pr.prove(None, seed=3)
        assert out["instances"] == O.fr_to_ints(np.stack(want))
        assert quotient_identity_holds(pr, out["challenges"], out["evals"], out["instances"])
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU))
        assert _verify(O, api, out["proof"], {**vk, "instances": out["instances"]})
        for i in range(4):                                                    # another distance than the circuit computed
            other = list(out["instances"])
            other[i] = (other[i] + 1) % O.R_MOD
            assert not _verify(O, api, out["proof"], {**vk, "instances": other})
        # other vectors under the same key
        rng = np.random.default_rng(9)
        v = rng.random((2, hp.dim))
        v[1, 1] = v[0, 1]
        hp.set_vectors(v)
        out2 = pr.prove(None)
        assert _verify(O, api, out2["proof"], {**vk, "instances": out2["instances"]}) and out2["instances"] != out["instances"]
        assert not _verify(O, api, out2["proof"], {**vk, "instances": out["instances"]})
        got = api.dequantize(hp.results())
        for m, g in zip(hp.metrics, got):
            f = E.F64[m](list(v[0]), list(v[1]))
            assert abs(float(g) - f) <= 1e-6 * max(abs(f), 1.0)
    finally:
        if pr is not None:
            pr.free()
        hp.free()


def test_c1_euclid_proves(api, O):
    """BASELINE configs[0] — one Euclidean distance of two 4-dim vectors, k = 13, LOOKUP_BITS = 12 — beyond its Mock stage
    (tests/test_gpu_mock.py): keygen, proof, verifier; and examples/euclid.rs's ten distances of one pair with nothing public"""
    from halo2_vectordb_amd.pipeline import DistancesHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import pairing as PR
    rng = np.random.default_rng(1)
    v = rng.random((2, 4))
    hp = DistancesHotPath(dim=4, metrics=("euclidean",), k=13, L=12, tau=TAU, vectors=v).setup()
    pr = ProverRounds(hp).keygen()
    try:
        assert pr.keygen_report.violations() == 0
        assert hp.n_adv_cols == 3 and hp.n_lk_cols == 1                       # 21.2 k / 3.4 k cells at L = 12
        out = pr.prove(None)
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU))
        assert len(out["instances"]) == 1 and _verify(O, api, out["proof"], {**vk, "instances": out["instances"]})
        f = float(np.linalg.norm(v[0] - v[1]))
        assert abs(float(api.dequantize(hp.results())[0]) - f) <= 1e-6 * max(f, 1.0)
    finally:
        pr.free()
        hp.free()
    d = E.load("euclid")
    r = E.oracle_distances(O, "euclid")
    hp = DistancesHotPath(dim=len(d["a"]), metrics=("euclidean",) * 10, k=13, L=12, tau=TAU, vectors=np.array([d["a"], d["b"]], dtype=np.float64), public=False).setup()
    pr = ProverRounds(hp).keygen()
    try:
        assert hp.n_cells == len(r["ctx"])
        _same_stream(hp, r["ctx"], O)
        assert pr.keygen_report.violations() == 0 and pr.instance_cells == []
        out = pr.prove(None)
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU))
        assert out["instances"] == [] and _verify(O, api, out["proof"], {**vk, "instances": []})
        assert E.rel_close(api.dequantize(hp.results()), [E.f64_euclidean(d["a"], d["b"])] * 10)
    finally:
        pr.free()
        hp.free()


def test_poseidon_example_proves_with_inputs_and_hash_public(api, O):
    """examples/poseidon.rs:26-36 on data/poseidon.in: two loaded field elements, update([x, y]); squeeze — merkle_commitment over ONE
    two-word vector is exactly that (one leaf, no tree level) —, x, y and the hash public (make_public.extend([x, y]); push(hash)).  The
    inputs are raw field elements here, not quantized numbers: the hot path's input buffer takes them as they are."""
    from halo2_vectordb_amd.pipeline import MerkleHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import pairing as PR
    r = E.oracle_poseidon(O)
    hp = MerkleHotPath(n=1, dim=2, k=10, tau=TAU).setup()
    pr = None
    try:
        hp.qvec = np.ascontiguousarray(r["xy"].reshape(1, 2, 4))
        hp.d_vec.upload(hp.qvec)
        pr = ProverRounds(hp).keygen()                       # the default statement: the hash alone
        root_cell = int(pr.root_cell)
        pr.free()
        pr = ProverRounds(hp).keygen(instance_cells=[0, 1, root_cell])
        assert pr.keygen_report.violations() == 0, pr.keygen_report.as_dict()
        out = pr.prove(None)
        x, y = O.fr_to_ints(r["xy"])
        h = O.fr_to_ints(np.asarray(r["hash"]).reshape(1, 4))[0]
        assert out["instances"] == [x, y, h] and (x, y) == (6, 100)
        assert np.array_equal(hp.results(), r["hash"])
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU))
        assert _verify(O, api, out["proof"], {**vk, "instances": out["instances"]})
        assert not _verify(O, api, out["proof"], {**vk, "instances": [x, y + 1, h]})
        assert not _verify(O, api, out["proof"], {**vk, "instances": [x, y, (h + 1) % O.R_MOD]})
    finally:
        if pr is not None:
            pr.free()
        hp.free()


@pytest.mark.parametrize("x", [1.128, -1.88724767676867])
def test_fixed_point_example_proves_with_x_and_its_results_public(api, O, x):
    """examples/fixed_point.rs:38-112 (FixedPointChip<32>; LOOKUP_BITS 12, k 13 as its commented-out set_var lines say): load_witness(x),
    qexp2, qlog2 when x > 0, qsin — the cells against the oracle's, the device MockProver on the whole map (qmod's asserted sign
    included), a proof whose instances are x and the results, the verifier; the proof bytes against the CPU prover's"""
    import math
    from halo2_vectordb_amd.pipeline import FixedPointHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from halo2_vectordb_amd import circuit_sym as CS
    from oracle import pairing as PR
    from oracle import prover as PV
    from test_gpu_cpu_prover import _compare
    P, L, k = 32, 12, 13
    hp = FixedPointHotPath(x=x, k=k, P=P, L=L, tau=TAU).setup()
    pr = None
    try:
        assert hp.ops == (("qexp2", "qlog2", "qsin") if x > 0 else ("qexp2", "qsin"))
        q = O.quantize(np.array([x]), P)
        c = O.Ctx(store=True, keygen=True, plan_k=k)
        c.assign_witnesses(q)
        res = [c.op(name, q[0], P=P, L=L) for name in hp.ops]
        assert hp.n_cells == len(c) and hp.n_lookup == c.n_lookup
        _same_stream(hp, c, O)
        assert np.array_equal(hp.results(), np.stack(res))
        f64 = dict(qexp2=lambda v: 2.0 ** v, qlog2=math.log2, qsin=math.sin)
        for name, got in zip(hp.ops, api.dequantize(hp.results(), P)):
            assert abs(float(got) - f64[name](x)) <= 5e-3, (name, float(got))        # 32 fractional bits (the example prints its errors)
        pr = ProverRounds(hp).keygen()
        assert pr.keygen_report.violations() == 0, pr.keygen_report.as_dict()
        assert len(pr.instance_cells) == 1 + len(hp.ops)
        cm, outs = CS.trace_fixed_point(hp.ops, P, L)
        cs = PV.Circuit(k, L, c.break_points(), c.selectors(), c.n_lookup, cm.copy_of, cm.const_idx, cm.consts, cm.lookup_src, [int(o) for o in outs])
        _pk, proofs = _compare(O, PV, hp, pr, cs, c.advice(), c.lookup(), seeds=(5,))
        out = proofs[0]
        assert out["instances"] == O.fr_to_ints(np.concatenate([q, np.stack(res)]))
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU))
        assert _verify(O, api, out["proof"], {**vk, "instances": out["instances"]})
        other = list(out["instances"])
        other[-1] = (other[-1] + 1) % O.R_MOD                                          # another sine than the circuit computed
        assert not _verify(O, api, out["proof"], {**vk, "instances": other})
    finally:
        if pr is not None:
            pr.free()
        hp.free()


@pytest.mark.parametrize("ops,x", [(("qcos", "qtanh"), 0.9), (("sign", "clip", "qsqrt", "qtan"), 2.75), (("neg", "qsinh", "qcosh", "qexp"), -1.5)])
def test_other_fixed_point_operations_prove(api, O, ops, x):
    """FixedPointHotPath with other operations than the example's three: the device MockProver on the traced map, a proof with x and the
    results public, the verifier — for the operations of the trait that nothing else in the suite proves"""
    from halo2_vectordb_amd.pipeline import FixedPointHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import pairing as PR
    P, L, k = 48, 12, 13
    hp = FixedPointHotPath(x=x, ops=ops, k=k, P=P, L=L, tau=TAU).setup()
    pr = None
    try:
        q = O.quantize(np.array([x]), P)
        c = O.Ctx(store=True, keygen=True, plan_k=k)
        c.assign_witnesses(q)
        res = [c.op(name, q[0], P=P, L=L) for name in ops]
        _same_stream(hp, c, O)
        assert np.array_equal(hp.results(), np.stack(res))
        pr = ProverRounds(hp).keygen()
        assert pr.keygen_report.violations() == 0, pr.keygen_report.as_dict()
        out = pr.prove(None)
        assert out["instances"] == O.fr_to_ints(np.concatenate([q, np.stack(res)]))
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU))
        assert _verify(O, api, out["proof"], {**vk, "instances": out["instances"]})
        other = list(out["instances"])
        other[1] = (other[1] + 1) % O.R_MOD
        assert not _verify(O, api, out["proof"], {**vk, "instances": other})
    finally:
        if pr is not None:
            pr.free()
        hp.free()
