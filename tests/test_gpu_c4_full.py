"""BASELINE configs[3] — the configuration the bench line is quoted on — at FULL size inside the driver-run GPU suite:
kmeans K=4 I=8 over 256 x 128 SIFT-shaped vectors, P=48, LOOKUP_BITS=15, k=16 (the reference's `chip_kmeans`,
/root/reference/tests/vectordb/mod.rs:93-135, and `examples/kmeans.rs:32-56`, at BASELINE size).

(a) the Euclidean circuit of the bench line (the reference tests' choice, tests/vectordb/mod.rs:109) through the resident hot path
    with the bench's seed: size, break points, results, the first k-means iteration's cells (66.7 M advice + lookup cells),
    sampled commitments and the coefficient / extended forms of two columns — all against the CPU oracle, bit for bit;
(b) the cosine circuit C4' (what examples/kmeans.rs:48-49 runs; the only satisfiable one) as a whole proof: the device
    MockProver with the circuit's whole constraint map and the 512 public centroid words reports nothing, one proof, the quotient
    identity at x recombined from its evaluations, and the pair (proof file, verifying-key file) through the stand-alone CPU
    verifier of tests/verify_file.py — accepted, and rejected with one byte changed.

Parity unpinned beyond the oracle: the reference records no outputs for this configuration."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(n=256, dim=128, K=4, I=8, k=16, P=48, L=15)        # bench.py's cfg
SEED = 20260004                                              # C4's seed (KmeansHotPath's default, the bench's)


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init(0)
    return a


def test_c4_euclidean_hot_path_against_the_oracle(api, O):
    from halo2_vectordb_amd.pipeline import KmeansHotPath, N_BLIND
    hp = KmeansHotPath(seed=SEED, blind_seed=77, **CFG).setup()      # (fixed blinds: the commitments are recomputed on the CPU)
    try:
        # the numbers DESIGN.md section 5 and the bench line quote
        assert (hp.n_cells, hp.n_lookup, hp.n_adv_cols, hp.n_lk_cols) == (445_579_138, 88_088_576, 6_801, 1_345)
        qv = O.quantize(hp.vectors_f64, CFG["P"])
        assert np.array_equal(qv, hp.qvec)
        # the whole circuit on the CPU without storing its cells: break points ("pinning") and results
        whole = O.Ctx(store=False, keygen=True, plan_k=CFG["k"])
        whole.assign_witnesses(qv)
        cent, ind = whole.kmeans("euclidean", qv, CFG["K"], CFG["I"], P=CFG["P"], L=CFG["L"])
        assert (len(whole), whole.n_lookup) == (hp.n_cells, hp.n_lookup)
        assert np.array_equal(whole.break_points(), hp.bp)
        del whole
        commitments = hp.step().copy()
        gc, gi = hp.results()
        assert np.array_equal(gc, cent) and np.array_equal(gi, ind)
        # every cluster keeps members through all eight iterations, and every vector has exactly one cluster
        one = O.fr_from_ints([1 << CFG["P"]])[0]
        assert ((gi == one).all(axis=2).sum(axis=1) == 1).all() and (gi == one).all(axis=2).any(axis=0).all()
        # the first iteration's cells: a prefix of the eight-iteration streams
        c = O.Ctx(store=True, keygen=False)
        c.assign_witnesses(qv)
        c.kmeans("euclidean", qv, CFG["K"], 1, P=CFG["P"], L=CFG["L"])
        adv, lk = c.advice(), c.lookup()
        del c
        assert len(adv) + len(lk) > 66_000_000
        assert np.array_equal(hp.d_stream.download((len(adv), 4)), adv)
        assert np.array_equal(hp.d_lookup.download((len(lk), 4)), lk)
        del adv, lk
        # coefficient and extended forms of the first two columns as the step left them
        coeff = hp.d_cols.download((2, hp.rows, 4))
        ext = hp.d_ext.download((2, 4 * hp.rows, 4))
        # 64 columns across the advice and lookup blocks, as laid out (blinding rows included): commitments
        hp.relayout()
        idx = sorted(set(list(range(0, hp.n_cols, hp.n_cols // 62))[:62] + [hp.n_adv_cols - 1, hp.n_cols - 1]))
        cols = hp.download_columns(idx)
        assert cols[:, hp.rows - N_BLIND:].any(axis=(1, 2)).all()          # blinded
        threads = min(16, len(os.sched_getaffinity(0)))
        assert np.array_equal(commitments[idx], O.msm_batch(cols, hp.g_lagrange, threads=threads))
        first = hp.download_columns([0, 1])
        want_coeff, want_ext = O.lde_batch(first, ext=2, threads=2)
        assert np.array_equal(coeff, want_coeff) and np.array_equal(ext, want_ext)
    finally:
        hp.free()
        from halo2_vectordb_amd._lib import check
        check(api.init().vdb_scratch_release())


def test_c4_cosine_whole_proof_mock_prover_and_stand_alone_verifier(api, O, tmp_path):
    from halo2_vectordb_amd.io import write_snark
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    hp = KmeansHotPath(seed=SEED, metric="cosine", **CFG)
    hp.ext_block_cols = 256          # as bench.py's whole_proof: the rounds recompute the cosets block by block
    hp.setup()
    pr = None
    try:
        assert hp.n_cols == 20_969 and hp.n_cells + hp.n_lookup > 1_370_000_000
        pr = ProverRounds(hp).keygen()
        # Mock stage on the keygen witness: gate rows, range table, 184 M copies, constants — nothing to report
        assert pr.keygen_report.violations() == 0, pr.keygen_report.as_dict()
        assert len(pr.instance_cells) == CFG["K"] * CFG["dim"] == 512
        out = pr.prove(None)
        inst = out["instances"]
        assert len(inst) == 512
        # the public centroid words are the witness's own (MockProver::run's instance argument) ...
        cent, _ind = hp.results()
        assert inst == [int(v) for v in O.fr_to_ints(cent.reshape(-1, 4))]
        rep = pr.mock_check_instances(inst)
        assert rep.violations() == 0, rep.as_dict()
        # ... and a claim of other centroids is reported
        wrong = list(inst)
        wrong[100] = (wrong[100] + 1) % O.R_MOD
        assert pr.mock_check_instances(wrong).violations() == 1
        assert quotient_identity_holds(pr, out["challenges"], out["evals"], inst)
        assert not quotient_identity_holds(pr, out["challenges"], out["evals"], wrong)
        path = str(tmp_path / "kmeans_c4_cosine.snark")
        write_snark(path, out["proof"], inst)
        pr.save_verifying_key(path + ".vk.npz", opened=out["opened"])
        n_proof = len(out["proof"])
        del out
    finally:
        if pr is not None:
            pr.free()
        hp.free()
        from halo2_vectordb_amd._lib import check
        check(api.init().vdb_scratch_release())
    # the stand-alone verifier: the two files and nothing else, on the CPU
    sys.path.insert(0, HERE)
    import verify_file
    rep = verify_file.main(path)
    assert rep["accepted"] and not rep["tampered_byte_accepted"], rep
    assert rep["columns"] == 20_969 and rep["proof_bytes"] == n_proof
