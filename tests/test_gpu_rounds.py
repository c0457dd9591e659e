"""The prover rounds after the advice commitments (halo2_vectordb_amd/rounds.py), end to end on a small k-means circuit:
everything a verifier would check, checked — the quotient identity at the evaluation point recombined from the returned
evaluations, and every opening against its commitments in the exponent (the test knows tau).  Parity unpinned: the
reference holds no vectors for the prover rounds; these are the PLONK / KZG identities themselves."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TAU = 0x1234567890ABCDEF1234567


@pytest.fixture(scope="module")
def proved(O):
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    api.init(0)
    hp = KmeansHotPath(n=8, dim=4, K=2, I=1, k=11, L=10, tau=TAU).setup()
    pr = ProverRounds(hp).keygen()
    rng = np.random.default_rng(99)
    ch = {name: O.random_fr(rng, 1)[0] for name in ("beta", "gamma", "y", "x", "v")}
    timings = {}
    out = pr.prove(ch, seed=5, timings=timings)
    yield hp, pr, ch, out, timings
    pr.free()
    hp.free()


def test_round_outputs_have_the_expected_shape(proved):
    hp, pr, ch, out, timings = proved
    assert pr.n_adv >= 2 and pr.n_lk >= 1
    c = out["commitments"]
    assert c["adv"].shape == (pr.n_cols, 8) and c["zp"].shape == (pr.n_sets, 8) and c["h"].shape == (4, 8) and c["pa"].shape == (pr.n_lk, 8)
    assert len(out["openings"]) == 6
    for name in ("witness", "commit_msm", "ntt", "lookup_permute", "products", "quotient", "evaluations", "openings"):
        assert timings[name] > 0


def test_quotient_identity_from_the_returned_evaluations(proved, O):
    from halo2_vectordb_amd.rounds import CHUNK_LEN, N_BLIND
    hp, pr, ch, out, _ = proved
    R = O.R_MOD
    to_int = lambda a: O.fr_to_ints(np.asarray(a).reshape(1, 4))[0]
    b, g, yv, x = (to_int(ch[n]) for n in ("beta", "gamma", "y", "x"))
    delta = to_int(pr.delta)
    ev = lambda name, rot=0: out["evals"][(name, rot)]
    n, n_adv = pr.rows, pr.n_adv
    acc = 0
    a0, a1, a2, a3, q = ev("adv"), ev("adv", 1), ev("adv", 2), ev("adv", 3), ev("sel")
    for c in range(n_adv):
        acc = (acc * yv + q[c] * (a0[c] + a1[c] * a2[c] - a3[c])) % R
    l0, ll, la = ev("lag")
    sg, z0, z1, zb = ev("sigma"), ev("zp"), ev("zp", 1), ev("zp", -N_BLIND)
    n_cols, n_sets = len(a0), len(z0)
    acc = (acc * yv + l0 * (1 - z0[0])) % R
    acc = (acc * yv + ll * (z0[-1] * z0[-1] - z0[-1])) % R
    for i in range(1, n_sets):
        acc = (acc * yv + l0 * (z0[i] - zb[i - 1])) % R
    cur = b * x % R
    for i in range(n_sets):
        left, right = z1[i], z0[i]
        for c in range(i * CHUNK_LEN, min((i + 1) * CHUNK_LEN, n_cols)):
            left = left * (a0[c] + b * sg[c] + g) % R
            right = right * (a0[c] + cur + g) % R
            cur = cur * delta % R
        acc = (acc * yv + la * (left - right)) % R
    A, S, PA, PS, PAm, Z, Z1 = a0[n_adv:], ev("table")[0], ev("pa"), ev("ps"), ev("pa", -1), ev("zl"), ev("zl", 1)
    for c in range(len(A)):
        acc = (acc * yv + l0 * (1 - Z[c])) % R
        acc = (acc * yv + ll * (Z[c] * Z[c] - Z[c])) % R
        acc = (acc * yv + la * (Z1[c] * (PA[c] + b) * (PS[c] + g) - Z[c] * (A[c] + b) * (S + g))) % R
        acc = (acc * yv + l0 * (PA[c] - PS[c])) % R
        acc = (acc * yv + la * (PA[c] - PS[c]) * (PA[c] - PAm[c])) % R
    xn = pow(x, n, R)
    hx = sum(h_i * pow(xn, i, R) for i, h_i in enumerate(ev("h"))) % R
    assert acc == hx * (xn - 1) % R and acc != 0


def test_every_opening_verifies_in_the_exponent(proved, O):
    """sum_i v^(m-1-i) C_i - [eval] G == [tau - point] W for each rotation point, with eval also equal to the same
    combination of the individual evaluations"""
    hp, pr, ch, out, _ = proved
    R = O.R_MOD
    v = O.fr_to_ints(ch["v"].reshape(1, 4))[0]
    G = O.g1_generator().reshape(1, 8)
    for op in out["openings"]:
        commits = np.concatenate([out["commitments"][name] for name in op["polys"]])
        evs = [e for name in op["polys"] for e in out["evals"][(name, op["rotation"])]]
        m = len(evs)
        assert commits.shape[0] == m
        comb = 0
        for e in evs:
            comb = (comb * v + e) % R
        assert comb == O.fr_to_ints(op["eval"].reshape(1, 4))[0]
        scalars = [pow(v, m - 1 - i, R) for i in range(m)] + [(-comb) % R]
        lhs = O.msm_naive(O.fr_from_ints(scalars), np.concatenate([commits, G]))
        rhs = O.msm_naive(O.fr_from_ints([(TAU - op["point"]) % R]), op["W"].reshape(1, 8))
        assert np.array_equal(lhs, rhs) and lhs.any()


def test_advice_commitments_are_the_hot_path_commitments(proved, O):
    hp, pr, ch, out, _ = proved
    hp.relayout()
    cols = hp.download_columns([0, pr.n_adv - 1, pr.n_adv])
    want = O.msm_batch(cols, hp.g_lagrange)
    assert np.array_equal(out["commitments"]["adv"][[0, pr.n_adv - 1, pr.n_adv]], want)
