"""Oracle pinning, part 1: BN254 constants, field/group arithmetic, MSM and NTT restatements.

No reference golden vectors exist for this layer (halo2curves/halo2-axiom are
un-vendored third-party crates) => pinned to the public BN254 constants of SURVEY App. D, to an
independent Python big-int implementation (oracle/pyref.py) and to algebraic invariants.
"""
import numpy as np

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def test_montgomery_constants(O):
    # SURVEY §8(b) [VERIFIED-HERE]
    assert O.R_MOD == R and O.Q_MOD == Q
    one = O.fr_from_canonical(O.ints_to_limbs([1]))
    assert O.limbs_to_ints(one)[0] == 0x0E0A77C19A07DF2F666EA36F7879462E36FC76959F60CD29AC96341C4FFFFFFB
    r2 = O.fr_from_canonical(one)  # R*R mod r, canonical view of mont(mont(1))
    assert O.limbs_to_ints(r2)[0] == 0x0216D0B17F4E44A58C49833D53BB808553FE3AB1E35C59E31BB8E645AE216DA7
    assert (-pow(R, -1, 1 << 64)) % (1 << 64) == 0xC2E1F593EFFFFFFF
    assert (-pow(Q, -1, 1 << 64)) % (1 << 64) == 0x87D20782E4866389


def test_root_of_unity_and_zeta(O):
    w = O.fr_to_ints(O.root_of_unity(28))[0]
    assert w == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    assert pow(w, 1 << 28, R) == 1 and pow(w, 1 << 27, R) != 1
    for k in (1, 5, 16, 18):
        wk = O.fr_to_ints(O.root_of_unity(k))[0]
        assert wk == pow(w, 1 << (28 - k), R)
    z = O.fr_to_ints(O.zeta())[0]
    assert z == 0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD
    assert pow(z, 3, R) == 1 and z != 1
    # the permutation argument's coset multiplier: halo2curves bn256 Fr::DELTA = GENERATOR^(2^S) with GENERATOR = 7, S = 28, as
    # published in the crate (fr.rs, recalled) — the value the oracle and the library compute from 7
    assert O.DELTA_INT == pow(7, 1 << 28, R) == 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2


def test_field_ops_vs_python(O):
    rng = np.random.default_rng(1)
    a = O.random_fr(rng, 200)
    b = O.random_fr(rng, 200)
    edge = [0, 1, R - 1, R - 2, 1 << 48, R - (1 << 97), (1 << 253), 2]
    a[: len(edge)] = O.fr_from_ints(edge)
    b[: len(edge)] = O.fr_from_ints(edge[::-1])
    ai, bi = O.fr_to_ints(a), O.fr_to_ints(b)
    assert O.fr_to_ints(O.fr_mul(a, b)) == [x * y % R for x, y in zip(ai, bi)]
    assert O.fr_to_ints(O.fr_add(a, b)) == [(x + y) % R for x, y in zip(ai, bi)]
    assert O.fr_to_ints(O.fr_sub(a, b)) == [(x - y) % R for x, y in zip(ai, bi)]
    assert O.fr_to_ints(O.fr_inv(a)) == [pow(x, R - 2, R) for x in ai]
    assert O.limbs_to_ints(O.fr_to_canonical(a)) == ai


def test_g1_and_msm_vs_python(O, PY):
    rng = np.random.default_rng(2)
    hs = [int(x) for x in rng.integers(1, 1 << 62, size=12)]
    bases = O.g1_mul_generator(hs)
    G = (1, 2)
    for h, b in zip(hs[:4], bases[:4]):
        x, y = O.fq_to_ints(b.reshape(2, 4))
        assert (x, y) == PY.g1_mul(G, h)
        assert (y * y - x * x * x - 3) % Q == 0
    sc = O.random_fr(rng, 12)
    sc[0] = 0
    sc[1] = O.fr_from_ints([1])[0]
    sc[2] = O.fr_from_ints([R - 1])[0]
    si = O.fr_to_ints(sc)
    want = PY.g1_mul(G, sum(s * h for s, h in zip(si, hs)) % R)
    for got in (O.msm_naive(sc, bases), O.msm(sc, bases, 1), O.msm(sc, bases, 3)):
        assert tuple(O.fq_to_ints(got.reshape(2, 4))) == want
    # identity result is (0,0)
    z = np.zeros((12, 4), dtype=np.uint64)
    assert not O.msm(z, bases).any()


def test_msm_pippenger_sizes(O, PY):
    rng = np.random.default_rng(3)
    for n in (1, 3, 5, 40, 300):
        hs = [int(x) for x in rng.integers(1, 1 << 62, size=n)]
        bases = O.g1_mul_generator(hs)
        sc = O.random_fr(rng, n)
        want = PY.g1_mul((1, 2), sum(s * h for s, h in zip(O.fr_to_ints(sc), hs)) % R)
        assert tuple(O.fq_to_ints(O.msm(sc, bases, 2).reshape(2, 4))) == want


def test_ntt_vs_naive_and_python(O, PY):
    rng = np.random.default_rng(4)
    for k in (1, 3, 6):
        n = 1 << k
        a = O.random_fr(rng, n)
        w = O.root_of_unity(k)
        got = O.ntt(a, w)
        assert np.array_equal(got, O.ntt_naive(a, w))
        if k <= 3:
            assert O.fr_to_ints(got) == PY.dft(O.fr_to_ints(a), O.fr_to_ints(w)[0])


def test_lagrange_to_coeff_and_extended(O):
    rng = np.random.default_rng(5)
    k = 5
    n = 1 << k
    evals = O.random_fr(rng, n)
    coeffs = O.lagrange_to_coeff(evals)
    # forward NTT of the coefficients returns the evaluations
    assert np.array_equal(O.ntt(coeffs, O.root_of_unity(k)), evals)
    ext = O.coeff_to_extended(coeffs, 2)
    ci = O.fr_to_ints(coeffs)
    z = O.fr_to_ints(O.zeta())[0]
    we = O.fr_to_ints(O.root_of_unity(k + 2))[0]
    # extended[i] = p(zeta * omega_ext^i)
    for i in (0, 1, 7, 4 * n - 1):
        x = z * pow(we, i, R) % R
        assert O.fr_to_ints(ext[i])[0] == sum(c * pow(x, j, R) for j, c in enumerate(ci)) % R


def test_srs_from_tau_commit_consistency(O, PY):
    """commit_lagrange(evals) == commit(coeffs): the same polynomial committed in both bases."""
    k = 4
    tau = 0x1234567
    g, gl = O.srs_from_tau(k, tau)
    rng = np.random.default_rng(6)
    evals = O.random_fr(rng, 1 << k)
    coeffs = O.lagrange_to_coeff(evals)
    a = O.msm(evals, gl)
    b = O.msm(coeffs, g)
    assert np.array_equal(a, b)
    p_tau = sum(c * pow(tau, j, R) for j, c in enumerate(O.fr_to_ints(coeffs))) % R
    assert tuple(O.fq_to_ints(a.reshape(2, 4))) == PY.g1_mul((1, 2), p_tau)


def test_prover_round_primitives_against_python(O):
    """orc_grand_product / orc_eval_poly (checkers of polyops.hip) against plain Python big-int arithmetic"""
    R = O.R_MOD
    rng = np.random.default_rng(99)
    n = 40
    num_i = [int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % R for _ in range(n)]
    den_i = [int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % R for _ in range(n)]
    den_i[17] = 0
    z = O.grand_product(O.fr_from_ints(num_i).reshape(1, n, 4), O.fr_from_ints(den_i).reshape(1, n, 4))
    want, acc = [], 1
    for i in range(n):
        want.append(acc)
        acc = acc * num_i[i] * (pow(den_i[i], -1, R) if den_i[i] else 0) % R
    assert O.fr_to_ints(z[0]) == want
    x = int(rng.integers(1, 1 << 62)) ** 3 % R
    got = O.eval_polys(O.fr_from_ints(num_i).reshape(1, n, 4), O.fr_from_ints([x])[0])
    assert O.fr_to_ints(got) == [sum(c * pow(x, i, R) for i, c in enumerate(num_i)) % R]


def test_pairing_self_checks():
    """oracle/pairing.py (test infrastructure for the verifier-side checks): the G2 generator lies on the twist and has order r,
    the pairing is bilinear and non-degenerate"""
    from oracle import pairing as PR
    assert PR.on_twist(PR.G2)
    assert PR.pt_mul(PR.G2, PR.R) is None and PR.pt_mul(PR.G1, PR.R) is None
    a, b = 0x1234567, 0xABCDEF0123
    aP, bQ = PR.pt_mul(PR.G1, a), PR.pt_mul(PR.G2, b)
    assert PR.on_twist(bQ)
    # e(aP, Q) e(-P, aQ) = 1;  e(aP, bQ) e(-abP, Q) = 1;  e(P, Q) != 1
    assert PR.pairing_product_is_one([(aP, PR.G2), (PR.pt_neg(PR.G1), PR.pt_mul(PR.G2, a))])
    assert PR.pairing_product_is_one([(aP, bQ), (PR.pt_neg(PR.pt_mul(PR.G1, a * b % PR.R)), PR.G2)])
    assert not PR.pairing_product_is_one([(PR.G1, PR.G2)])
    assert not PR.pairing_product_is_one([(aP, bQ), (PR.pt_neg(PR.pt_mul(PR.G1, a * b + 1)), PR.G2)])
