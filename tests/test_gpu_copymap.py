"""The fixed-point circuits with their whole constraint map (halo2_vectordb_amd/circuit_sym.py) on GPU witnesses and through
the prover rounds: what the reference's Mock arm (MockProver) and Prove arm (create_proof's permutation argument over every
copy halo2-base records, src/scaffold/mod.rs:263-266, 296) enforce.

* device MockProver with the map: GPU witnesses of distance / nearest_vector / k-means satisfy every gate, copy, constant and
  lookup tie; Euclidean k-means violates only qlog2's asserted constants at iteration 0 and its proof is rejected;
* a witness whose copy of a qmul result was altered (gates repaired around it) proves without the map and is rejected with it;
* BASELINE configs[1] (nearest_vector over 64 x 128 at k = 14) proves and verifies with the whole map."""
import numpy as np
import pytest

from test_gpu_rounds import FIXED, TAU, _meta, _verify

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init(0)
    return a


def table(O, consts):
    return O.fr_from_ints(consts) if len(consts) else np.zeros((1, 4), dtype=np.uint64)


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "manhattan", "hamming"])
def test_distance_witness_against_the_symbolic_map(api, O, metric):
    """examples/distances.rs shape (two assigned vectors, one distance), P = 48, LOOKUP_BITS = 12: the GPU's cells under the
    whole map, on the device checker"""
    from halo2_vectordb_amd import circuit_sym as CS
    dim = 4
    cm, outs = CS.trace_distance(metric, dim, 48, 12)
    rng = np.random.default_rng(3)
    for a, b in ((rng.uniform(-2, 2, dim), rng.uniform(-2, 2, dim)), ([0.123, 0.456, 1.789, 1.123], [1.123, 0.456, 0.789, 0.123])):
        qa, qb = api.quantize([a]), api.quantize([b])
        got = api.wit_distance(metric, qa, qb, L=12, selectors=True)
        stream = np.concatenate([qa[0], qb[0], got["stream"]])
        flags = np.concatenate([np.zeros(2 * dim, dtype=np.uint8), got["flags"]])
        assert stream.shape[0] == cm.n_cells and np.array_equal((flags & 1).astype(bool), cm.gate)
        assert (cm.const_idx[(flags & 2) != 0] >= 0).all()          # every cell the kernels flag constant is a constant of the map
        bufs = [api.DeviceBuffer(max(x.nbytes, 32)) for x in (stream, flags, got["lookup"], cm.copy_of, cm.lookup_src, cm.const_idx, table(O, cm.consts))]
        for b_, x in zip(bufs, (stream, flags, got["lookup"], cm.copy_of, cm.lookup_src, cm.const_idx, table(O, cm.consts))):
            b_.upload(np.ascontiguousarray(x))
        rep = api.mock_check_dev(bufs[0].ptr, cm.n_cells, bufs[1].ptr, bufs[2].ptr, len(cm.lookup_src), 12, bufs[3].ptr, bufs[4].ptr, None, bufs[5].ptr, bufs[6].ptr,
                                 len(cm.consts))
        for b_ in bufs:
            b_.free()
        assert rep.violations() == 0, rep.as_dict()
        assert np.array_equal(stream[outs[0]], got["result"][0])


def test_kmeans_cosine_is_satisfied_and_euclidean_is_not(api, O):
    """examples/kmeans.rs:48-49: "until I can solve the bug with euclidean, we are not using Euclidean distance".  With every
    constraint in place the cosine circuit is satisfied; the Euclidean one violates exactly the asserted constants of qlog2 at
    iteration 0 (each initial centroid is its own vector: qsqrt(0)) and nothing else, and its proof does not verify."""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    from oracle import pairing as PR
    out = {}
    for metric in ("cosine", "euclidean"):
        hp = KmeansHotPath(n=8, dim=4, K=2, I=2, k=12, L=11, metric=metric, tau=TAU).setup()
        pr = ProverRounds(hp).keygen()
        try:
            rep = pr.keygen_report
            proof = pr.prove(None, seed=3)
            vk = dict(meta=_meta(pr), opened=proof["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU),
                      instances=proof["instances"])
            out[metric] = (rep.as_dict(), quotient_identity_holds(pr, proof["challenges"], proof["evals"], proof["instances"]), _verify(O, api, proof["proof"], vk))
            # another database of the same shape under the same key
            rng = np.random.default_rng(12)
            hp.set_vectors(rng.integers(0, 219, size=(8, 4)).astype(np.float64) + rng.random((8, 4)))
            rep2 = pr.mock_check()
            assert (rep2.violations() == 0) == (metric == "cosine"), rep2.as_dict()
        finally:
            pr.free()
            hp.free()
    rep, ident, ok = out["cosine"]
    assert sum(v for k, v in rep.items() if not k.startswith("first")) == 0 and ident and ok
    rep, ident, ok = out["euclidean"]
    assert rep["constants_changed"] > 0 and rep["gate_rows_violated"] == rep["copies_unequal"] == rep["lookup_copies_unequal"] == rep["lookup_cells_out_of_table"] == 0
    assert not ident and not ok


def test_an_altered_copy_of_a_qmul_result_is_rejected(api, O):
    """The cheating prover the permutation argument is there for: in nearest_vector's first inner product, the cell of
    `res = qadd(res, a_i b_i)` that copies the qmul's result is given another value and the gate's output is recomputed, so every
    gate row still holds and every lookup cell is still in the table.  Without the gadget's copy map the proof verifies; with it
    the device checker reports the broken copies and the verifier rejects."""
    from halo2_vectordb_amd.circuit_sym import CopyMap
    from halo2_vectordb_amd.pipeline import NearestHotPath
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    from oracle import pairing as PR
    hp = NearestHotPath(n=6, dim=4, k=12, L=11, tau=TAU).setup()
    pr = ProverRounds(hp).keygen()
    cm = pr.circuit
    # a gate [res, a_i b_i, 1, out]: operand cell (row + 1) copies an earlier cell, row + 2 is the constant one, out is fresh
    rows = np.flatnonzero(cm.gate)
    cand = [r for r in rows[200:] if cm.copy_of[r + 1] != r + 1 and cm.const_idx[r + 2] >= 0 and cm.consts[cm.const_idx[r + 2]] == 1
            and cm.copy_of[r + 3] == r + 3 and cm.const_idx[r + 3] < 0 and cm.const_idx[r + 1] < 0]
    row = int(cand[0])
    honest = hp._witness
    one = O.fr_from_ints([1])

    def tampered(sel=None):
        honest(sel)
        if sel is None:
            for cell in (row + 1, row + 3):           # operand + 1 and output + 1: the row still satisfies a + b * 1 = d
                v = hp.d_stream.download((1, 4), offset=cell * 32)
                hp.d_stream.upload(O.fr_add(v, one), offset=cell * 32)

    tau_h = PR.pt_mul(PR.G2, TAU)
    loose = ProverRounds(hp).keygen(circuit=CopyMap(np.arange(hp.n_cells, dtype=np.int64), cm.const_idx, cm.consts, cm.asserted, cm.gate, cm.lookup_src))
    results = {}
    try:
        for label, p in (("whole map", pr), ("without the gadget's copies", loose)):
            vk = lambda o: dict(meta=_meta(p), opened=o["opened"], fixed={name: p.fixed[name].commits for name in FIXED}, tau_h=tau_h, instances=o["instances"])
            good = p.prove(None, seed=5)
            assert _verify(O, api, good["proof"], vk(good)), label
            hp._witness = tampered
            try:
                bad = p.prove(None, seed=5)
                d_flags = api.DeviceBuffer(hp.n_cells)
                d_flags.upload(cm.gate.astype(np.uint8))
                rep = p.mock_check(d_flags)          # the tampered witness is what lies in HBM now
                d_flags.free()
            finally:
                hp._witness = honest
            results[label] = (rep.gate_rows_violated, rep.lookup_cells_out_of_table, rep.copies_unequal > 0,
                              quotient_identity_holds(p, bad["challenges"], bad["evals"], bad["instances"]), _verify(O, api, bad["proof"], vk(bad)))
    finally:
        loose.free()
        pr.free()
        hp.free()
    assert results["whole map"] == (0, 0, True, False, False)
    assert results["without the gadget's copies"] == (0, 0, False, True, True)


def test_c2_nearest_64x128_proves_with_the_whole_map(api, O):
    """BASELINE configs[1]: nearest_vector query over 64 x 128-dim SIFT-shaped vectors, k = 14, a real KZG proof on one MI355X:
    the whole constraint map in the permutation argument, proof bytes accepted by the stand-alone verifier (pairing check)"""
    from halo2_vectordb_amd.pipeline import NearestHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import pairing as PR
    hp = NearestHotPath(n=64, dim=128, k=14, L=13, tau=TAU).setup()
    pr = ProverRounds(hp).keygen()
    try:
        assert pr.keygen_report.violations() == 0, pr.keygen_report.as_dict()
        cm = pr.circuit
        tied = (cm.copy_of != np.arange(cm.n_cells)) | (cm.const_idx >= 0)
        assert cm.n_cells == hp.n_cells and tied.mean() > 0.5
        out = pr.prove(None)
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU),
                  instances=out["instances"])
        assert _verify(O, api, out["proof"], vk)
        bad = bytearray(out["proof"])
        bad[40] ^= 1
        assert not _verify(O, api, bytes(bad), vk)
        # the statement: the nearest vector's 128 words are public (examples/query.rs:58), tied to the instance column
        _ind, res = hp.results()
        assert out["instances"] == O.fr_to_ints(res) and len(out["instances"]) == 128
        other = list(out["instances"])
        other[77] = (other[77] + 1) % O.R_MOD
        assert not _verify(O, api, out["proof"], {**vk, "instances": other})
    finally:
        pr.free()
        hp.free()


def test_public_instances_bind_the_statement(api, O):
    """What RangeWithInstanceCircuitBuilder adds to the reference's circuits (src/scaffold/mod.rs:400, :265): the cells the closure
    pushes into make_public — the K x dim centroid words of examples/kmeans.rs:51-56 — are tied to an instance column.  An honest
    proof verifies against the centroids the witness holds and against nothing else; the device MockProver takes the instances
    (MockProver::run's third argument); a prover that claims other centroids than its witness computes cannot prove them."""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    from oracle import pairing as PR
    hp = KmeansHotPath(n=8, dim=4, K=2, I=2, k=12, L=11, metric="cosine", tau=TAU).setup()
    pr = ProverRounds(hp).keygen()
    try:
        assert len(pr.instance_cells) == 2 * 4 and pr.n_perm == pr.n_cols + 2
        out = pr.prove(None, seed=21)
        cent, _ind = hp.results()
        want = O.fr_to_ints(cent.reshape(-1, 4))
        assert out["instances"] == want                                   # row-major: every centroid, word by word
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU))
        assert "inst" not in vk["fixed"] and all(name != "inst" for names in out["opened"].values() for name in names)
        assert _verify(O, api, out["proof"], {**vk, "instances": want})
        for i in (0, 5, 7):                                               # a changed centroid word: rejected
            other = list(want)
            other[i] = (other[i] + 1) % O.R_MOD
            assert not _verify(O, api, out["proof"], {**vk, "instances": other})
        assert not _verify(O, api, out["proof"], {**vk, "instances": want[:-1]})
        assert not _verify(O, api, out["proof"], {**vk, "instances": want[1:] + want[:1]})      # the order is part of the statement
        # the Mock stage with instances
        assert pr.mock_check(instances=want).violations() == 0
        other = list(want)
        other[5] = (other[5] + 3) % O.R_MOD
        rep = pr.mock_check(instances=other)
        assert rep.instances_unequal == 1 and rep.first_instance == 5 and rep.violations() == 1
        # a prover that states centroids its witness does not compute: the instance column no longer equals the public cells
        lie = pr.prove(None, seed=22, instances=[O.fr_from_ints([v])[0] for v in other])
        assert lie["instances"] == other
        assert not quotient_identity_holds(pr, lie["challenges"], lie["evals"], other)
        assert not _verify(O, api, lie["proof"], {**vk, "opened": lie["opened"], "instances": other})
        assert not _verify(O, api, lie["proof"], {**vk, "opened": lie["opened"], "instances": want})
        # and an honest proof again (the lie left nothing behind in the instance column)
        again = pr.prove(None, seed=23)
        assert _verify(O, api, again["proof"], {**vk, "opened": again["opened"], "instances": want})
    finally:
        pr.free()
        hp.free()


def test_device_permutation_mapping_has_the_cycles_of_the_host_construction(api, O):
    """vdb_permutation_mapping_dev (pointer jumping + radix sort on the device) against copymap.mapping_from_copy_of (numpy): both
    are permutations of the (n_cols + 2) x rows grid ([advice | lookup | constants | instance]) with exactly the same classes —
    the order inside a cycle is free; the rows of the instance column sit in the classes of the public cells"""
    from halo2_vectordb_amd.copymap import mapping_from_copy_of
    from halo2_vectordb_amd.pipeline import MINIMUM_ROWS, KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    hp = KmeansHotPath(n=6, dim=3, K=2, I=1, k=11, L=10, metric="cosine", tau=TAU).setup()
    pr = ProverRounds(hp)
    pr.keep_mapping = True
    pr.keygen()
    try:
        cm, rows, n_perm = pr.circuit, pr.rows, pr.n_perm
        dev = pr._d_map_for_tests.download((n_perm, rows), dtype=np.uint64)
        host = mapping_from_copy_of(cm.copy_of, hp.bp, pr.n_cols, rows, cm.lookup_src, rows - MINIMUM_ROWS, const_idx=cm.const_idx, n_consts=len(cm.consts),
                                    instance_cells=pr.instance_cells)
        assert len(pr.instance_cells) == 2 * 3 and dev.shape == host.shape

        def classes(mapping):
            nxt = ((mapping >> np.uint64(32)).astype(np.int64) * rows + (mapping & np.uint64(0xFFFFFFFF)).astype(np.int64)).reshape(-1)
            assert np.array_equal(np.sort(nxt), np.arange(nxt.size))              # a permutation
            label = np.arange(nxt.size)
            while True:                                                            # smallest index of each cycle, by pointer doubling
                new = np.minimum(label, label[nxt])
                nxt = nxt[nxt]
                if np.array_equal(new, label):
                    return label
                label = new
        assert np.array_equal(classes(dev), classes(host))
        # row i of the instance column and the i-th public cell's grid position are in one class
        starts = np.concatenate([[0], np.cumsum(np.asarray(hp.bp, dtype=np.int64))])
        lab = classes(dev)
        for i, cell in enumerate(pr.instance_cells):
            col = int(np.searchsorted(starts, cell, side="right") - 1)
            assert lab[(pr.n_cols + 1) * rows + i] == lab[col * rows + cell - int(starts[col])]
        assert (classes(dev) != np.arange(n_perm * rows)).mean() > 0.3
    finally:
        pr._d_map_for_tests.free()
        pr.free()
        hp.free()


def test_streamed_rounds_give_the_resident_proof(api, O):
    """A circuit whose extended cosets do not fit HBM proves by streaming: the advice cosets are recomputed a block of columns at
    a time inside the quotient, the permutation's Lagrange and sigma columns and every derived coset exist one block at a time.
    Forced here on a small circuit (7 coset columns held, blocks of 12 columns): byte-identical proof to the resident run."""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import pairing as PR
    cfg = dict(n=8, dim=4, K=2, I=1, k=12, L=11, metric="cosine", tau=TAU)
    proofs = {}
    for label, ext_block, block_cols in (("resident", None, 510), ("streamed", 7, 12)):
        hp = KmeansHotPath(**cfg)
        hp.ext_block_cols = ext_block
        hp.setup()
        pr = ProverRounds(hp, block_cols=block_cols).keygen()
        try:
            assert (hp.ext_cols >= hp.n_cols + 2) == (label == "resident") and pr.n_perm > 5 * 12
            out = pr.prove(None, seed=9)
            vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU),
                      instances=out["instances"])
            assert _verify(O, api, out["proof"], vk), label
            proofs[label] = out["proof"]
            # the sigma and selector cosets stay in HBM when they fit (they do here); made block by block inside the quotient instead
            # — what a circuit the size of BASELINE C4' does on one card — the proof is the same, byte for byte
            assert pr.fixed_cosets_resident and pr.fixed["sigma"].ext is not None
            pr.fixed_cosets_resident = False
            assert pr.prove(None, seed=9)["proof"] == out["proof"], label
            pr.fixed_cosets_resident = True
        finally:
            pr.free()
            hp.free()
    assert proofs["resident"] == proofs["streamed"]


@pytest.mark.parametrize("shape", [("kmeans", "cosine", 9, 5, 3, 2), ("kmeans", "euclidean", 20, 6, 4, 3), ("nearest", "euclidean", 7, 4, 0, 0), ("nearest", "cosine", 12, 3, 0, 0)])
def test_constraint_map_built_on_the_device_is_the_host_built_map(api, O, shape):
    """circuit_dev.DeviceBuilder (vdb_copymap_place_dev: one kernel per placement of a traced unit block) against the numpy
    builder of circuit_sym.py that tests/test_circuit_sym_cpu.py holds against real witnesses: the same copy_of, const_idx, gate
    and asserted flags, lookup sources, constants and output cells; keygen's parent array from the device map closes the same way"""
    import ctypes
    from halo2_vectordb_amd import circuit_sym as CS
    from halo2_vectordb_amd.circuit_dev import DeviceBuilder
    kind, metric, n, dim, K, I = shape
    if kind == "kmeans":
        host, outs_h = CS.build_kmeans(metric, n, dim, K, I, 48, 10)
        dev, outs_d = CS.build_kmeans(metric, n, dim, K, I, 48, 10, builder=DeviceBuilder)
    else:
        host, outs_h = CS.build_nearest(metric, n, dim, 48, 10)
        dev, outs_d = CS.build_nearest(metric, n, dim, 48, 10, builder=DeviceBuilder)
    try:
        assert dev.n_cells == host.n_cells and dev.consts == host.consts
        for a, b in zip(outs_h, outs_d):
            assert np.array_equal(np.asarray(a), np.asarray(b))
        for name in ("copy_of", "const_idx", "gate", "asserted", "lookup_src"):
            assert np.array_equal(getattr(dev, name), getattr(host, name)), name
        # the parent array keygen hands to vdb_permutation_mapping_dev
        d_parent = api.DeviceBuffer(host.n_cells * 8)
        bad, nosrc = ctypes.c_uint64(), ctypes.c_uint64()
        lib = api.init()
        assert lib.vdb_copymap_finish_dev(dev.d_copy_of.ptr, dev.d_const_idx.ptr, ctypes.c_uint64(host.n_cells), dev.d_lookup_src.ptr, ctypes.c_uint64(len(host.lookup_src)),
                                          d_parent.ptr, ctypes.byref(bad), ctypes.byref(nosrc)) == 0
        want = host.copy_of.astype(np.int64, copy=True)
        tied = host.const_idx >= 0
        want[tied] = host.n_cells + host.const_idx[tied]
        assert bad.value == 0 and nosrc.value == 0 and np.array_equal(d_parent.download((host.n_cells,), dtype=np.int64), want)
        d_parent.free()
    finally:
        dev.free()
