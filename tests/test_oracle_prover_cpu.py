"""The oracle's CPU prover (oracle/prover.py: `create_proof` composed from the oracle's bricks, test infrastructure) on its own, no GPU:
BASELINE C1 — euclidean_distance on two 4-dim vectors, LOOKUP_BITS = 12, k = 13 — is proved on the CPU and the proof bytes go through
the stand-alone verifier of tests/test_gpu_rounds.py (`_verify`: transcript replayed by the PRODUCT library's host sponge — so the
prover's Python-integer sponge and the library's C++ one are held to each other here —, quotient identity, the SHPLONK pairing
equation).  The GPU prover is held to this prover byte for byte in tests/test_gpu_cpu_prover.py; the committed digest below pins the
bytes themselves across rounds (tests/golden/cpu_prover_golden.json, written by this file run as a script)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
GOLDEN = os.path.join(HERE, "golden", "cpu_prover_golden.json")
TAU = 0x1234567890ABCDEF1234567
A, B = [0.123, 0.456, 1.789, 1.123], [1.123, 0.456, 0.789, 0.123]      # data/distances.in extended to 4 dims
K, P, L, SEED = 13, 48, 12, 13


def _c1(O, PV):
    from halo2_vectordb_amd import circuit_sym as CS          # the gadgets' symbolic trace: host Python, no device
    qa, qb = O.quantize(A, P), O.quantize(B, P)
    c = O.Ctx(store=True, keygen=True, plan_k=K)
    c.assign_witnesses(qa)
    c.assign_witnesses(qb)
    dist = c.distance("euclidean", qa, qb, P=P, L=L)
    cm, outs = CS.trace_distances(("euclidean",), 4, P, L)
    cs = PV.Circuit(K, L, c.break_points(), c.selectors(), c.n_lookup, cm.copy_of, cm.const_idx, cm.consts, cm.lookup_src, [int(x) for x in outs])
    return cs, c.advice(), c.lookup(), dist


def _prove(O, PV):
    cs, stream, lookup, dist = _c1(O, PV)
    g, gl = O.srs_from_tau(K, TAU)
    pk = PV.keygen(cs, g, gl, threads=4)
    out = PV.prove(pk, stream, lookup, PV.seeded_blinds(cs, SEED))
    return cs, pk, out, stream, lookup, dist


def _vk(O, cs, pk, out):
    from oracle import pairing as PR
    meta = dict(rows=cs.rows, k=cs.k, n_adv=cs.n_adv, n_lk=cs.n_lk, n_cols=cs.n_cols, n_sets=cs.n_sets, chunk_len=cs.chunk_len, n_blind=PV_N_BLIND,
                delta=O.DELTA_INT, n_instances=len(cs.instance_cells))
    return dict(meta=meta, opened=out["opened"], fixed={name: pk.commits[name] for name in ("sel", "sigma", "cst", "table")},
                tau_h=PR.pt_mul(PR.G2, TAU), instances=out["instances"])


PV_N_BLIND = 7


def test_cpu_prover_c1_verifies_and_keeps_its_bytes(O):
    from halo2_vectordb_amd import api
    from oracle import prover as PV
    from test_gpu_rounds import _verify
    cs, pk, out, stream, lookup, dist = _prove(O, PV)
    assert (cs.n_adv, cs.n_lk, cs.degree, cs.chunk_len, cs.n_h) == (3, 1, 4, 2, 3)
    assert out["instances"] == O.fr_to_ints(dist.reshape(1, 4))
    vk = _vk(O, cs, pk, out)
    assert _verify(O, api, out["proof"], vk)
    bad = bytearray(out["proof"])
    bad[len(bad) // 2] ^= 1
    assert not _verify(O, api, bytes(bad), vk)
    assert not _verify(O, api, out["proof"], {**vk, "instances": [(out["instances"][0] + 1) % O.R_MOD]})
    # the key's digest as the product's host sponge computes it
    tr = api.Transcript()
    for name in ("sel", "sigma", "cst", "table"):
        tr.common_points(pk.commits[name])
    assert O.fr_to_ints(tr.squeeze().reshape(1, 4))[0] == pk.vk_digest
    tr.free()
    G = json.load(open(GOLDEN))
    assert hashlib.sha256(out["proof"]).hexdigest() == G["c1_proof_sha256"] and len(out["proof"]) == G["c1_proof_bytes"]
    assert "%064x" % pk.vk_digest == G["c1_vk_digest"]
    # a witness that breaks a gate has no quotient of degree below (degree - 1) n
    broken = stream.copy()
    broken[20] = O.fr_add(broken[20:21], O.fr_from_ints([1]))[0]
    try:
        PV.prove(pk, broken, lookup, PV.seeded_blinds(cs, SEED))
        raise AssertionError("an unsatisfied circuit was proved")
    except ValueError:
        pass


def test_permutation_mapping_is_a_permutation_with_the_copy_classes_as_cycles(O):
    from oracle import prover as PV
    cs, stream, lookup, _ = _c1(O, PV)
    m = PV.permutation_mapping(cs)
    flat = (m >> np.uint64(32)).astype(np.int64) * cs.rows + (m & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.array_equal(np.sort(flat.reshape(-1)), np.arange(cs.n_perm * cs.rows))          # a permutation of the grid
    # every cycle holds one value: walk the grid values through the mapping
    inst = np.zeros((1, cs.rows, 4), dtype=np.uint64)
    inst[0, 0] = stream[cs.instance_cells[0]]
    cst = np.zeros((1, cs.rows, 4), dtype=np.uint64)
    cst[0, : len(cs.consts)] = O.fr_from_ints(cs.consts)
    grid = np.concatenate([PV.layout_advice(cs, stream), PV.layout_lookup(cs, lookup), cst, inst]).reshape(-1, 4)
    assert np.array_equal(grid[flat.reshape(-1)], grid)
    moved = flat.reshape(-1) != np.arange(flat.size)
    assert moved.sum() > cs.n_lookup                                                         # lookup cells, copies, constants are tied


if __name__ == "__main__":
    from oracle import oracle as O_, prover as PV_
    _, pk_, out_, *_ = _prove(O_, PV_)
    json.dump({"circuit": "BASELINE C1: euclidean_distance, dim 4, P=48, LOOKUP_BITS=12, k=13, tau=TAU, blinds seeded_blinds(13)",
               "c1_proof_sha256": hashlib.sha256(out_["proof"]).hexdigest(), "c1_proof_bytes": len(out_["proof"]), "c1_vk_digest": "%064x" % pk_.vk_digest},
              open(GOLDEN, "w"), indent=1)
    print(open(GOLDEN).read())
