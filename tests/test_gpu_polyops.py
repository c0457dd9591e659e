"""Grand product of the permutation / lookup arguments (first brick) — GPU vs the CPU restatement, bit exact,
plus the defining recurrence at the bench size.  Parity unpinned beyond the oracle: the reference holds no vectors for the
prover rounds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init(0)
    return a


@pytest.mark.parametrize("n_cols,n", [(1, 1), (1, 2), (3, 7), (2, 1024), (2, 1025), (3, 4096), (1, 65536)])
def test_grand_product_matches_oracle(api, O, n_cols, n):
    rng = np.random.default_rng(31 * n + n_cols)
    num = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    den = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    assert np.array_equal(api.grand_product(num, den), O.grand_product(num, den))


def test_grand_product_zero_denominators_and_identity(api, O):
    rng = np.random.default_rng(77)
    n = 3000
    num = O.random_fr(rng, 3 * n).reshape(3, n, 4)
    den = O.random_fr(rng, 3 * n).reshape(3, n, 4)
    den[0, 1234] = 0                    # z is zero from row 1235 on (batch_invert leaves a zero denominator at zero)
    den[1, 0] = 0
    den[1, 2999] = 0                    # the last row never enters the product
    num[2] = den[2]                     # ratios of one: z is identically one (a satisfied permutation argument)
    z = api.grand_product(num, den)
    assert np.array_equal(z, O.grand_product(num, den))
    one = O.fr_from_ints([1])[0]
    assert np.array_equal(z[0, 1234], O.fr_mul(O.fr_mul(z[0, 1233].reshape(1, 4), num[0, 1233].reshape(1, 4)), O.fr_inv(den[0, 1233].reshape(1, 4)))[0])
    assert not z[0, 1235:].any() and not z[1, 1:].any()
    assert all(np.array_equal(r, one) for r in z[2][:: 97])


def test_grand_product_recurrence_at_bench_size(api, O):
    """size-independent property, 64 columns of 2^16: z[i+1] * den[i] == z[i] * num[i] and z[0] == 1"""
    rng = np.random.default_rng(5)
    n_cols, n = 64, 1 << 16
    num = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    den = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    z = api.grand_product(num, den)
    lhs = api.fr_mul(z[:, 1:].reshape(-1, 4), den[:, :-1].reshape(-1, 4))
    rhs = api.fr_mul(z[:, :-1].reshape(-1, 4), num[:, :-1].reshape(-1, 4))
    assert np.array_equal(lhs, rhs)
    assert np.array_equal(z[:, 0], np.tile(O.fr_from_ints([1]), (n_cols, 1)))
    assert np.array_equal(z[:2], O.grand_product(num[:2], den[:2]))


@pytest.mark.parametrize("n_cols,n", [(1, 1), (2, 5), (3, 255), (2, 256), (2, 257), (3, 1024), (2, 65536)])
def test_eval_polys_matches_oracle(api, O, n_cols, n):
    rng = np.random.default_rng(17 * n + n_cols)
    coeffs = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    for x in (O.random_fr(rng, 1)[0], O.fr_from_ints([0])[0], O.fr_from_ints([1])[0]):
        assert np.array_equal(api.eval_polys(coeffs, x), O.eval_polys(coeffs, x))


def test_eval_of_interpolant_returns_the_lagrange_values(api, O):
    """ties the NTT and the evaluation together at the bench size: the coefficients lagrange_to_coeff produces, evaluated
    at omega^j, are the Lagrange values the column started from"""
    rng = np.random.default_rng(23)
    k = 16
    cols = O.random_fr(rng, 3 << k).reshape(3, 1 << k, 4)
    coeffs = api.lagrange_to_coeff(cols)
    w = O.root_of_unity(k)
    for j in (0, 1, 2, 12345, (1 << k) - 1):
        x = O.fr_from_ints([pow(O.fr_to_ints(w.reshape(1, 4))[0], j, O.R_MOD)])[0]
        assert np.array_equal(api.eval_polys(coeffs, x), cols[:, j])
