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


def _range_table(O, n, bits):
    """halo2-base range table column: 0 .. 2^bits - 1 in the first rows, zero padding below"""
    return O.fr_from_ints([i if i < (1 << bits) else 0 for i in range(n)])


@pytest.mark.parametrize("n,usable,bits,n_cols", [(64, 58, 4, 2), (1024, 1018, 8, 3), (4096, 4090, 11, 2), (65536, 65530, 15, 2)])
def test_lookup_permute_matches_oracle(api, O, n, usable, bits, n_cols):
    rng = np.random.default_rng(41 * bits + n_cols)
    table = _range_table(O, n, bits)
    ins = []
    for c in range(n_cols):
        v = rng.integers(0, 1 << bits, size=n)
        if c == 1:
            v[: n // 2] = 0          # long run of zeros (the padding cells of a lookup column)
        ins.append(O.fr_from_ints([int(x) for x in v]))
    ins = np.stack(ins)
    got_a, got_s = api.lookup_permute(ins, table, usable, bits)
    tab_ints = O.fr_to_ints(table[:usable])
    for c in range(n_cols):
        want_a, want_s = O.lookup_permute(O.fr_to_ints(ins[c, :usable]), tab_ints)
        assert O.fr_to_ints(got_a[c, :usable]) == want_a
        assert O.fr_to_ints(got_s[c, :usable]) == want_s
        assert not got_a[c, usable:].any() and not got_s[c, usable:].any()
        # what the lookup argument needs: permutations, and A'[i] in {S'[i], A'[i-1]}
        assert sorted(want_s) == sorted(tab_ints)
        assert all(want_a[i] == want_s[i] or (i and want_a[i] == want_a[i - 1]) for i in range(usable))


def test_lookup_permute_errors(api, O):
    n, usable, bits = 256, 250, 6
    table = _range_table(O, n, bits)
    ok = O.fr_from_ints([i % 64 for i in range(n)]).reshape(1, n, 4)
    api.lookup_permute(ok, table, usable, bits)
    bad = ok.copy()
    bad[0, 7] = O.fr_from_ints([64])[0]                 # not below 2^bits
    with pytest.raises(api.VdbError) as e:
        api.lookup_permute(bad, table, usable, bits)
    assert e.value.code == -5
    short_table = O.fr_from_ints([i if i < 32 else 0 for i in range(n)])   # 40 is not in this table
    with pytest.raises(api.VdbError) as e:
        api.lookup_permute(ok, short_table, usable, bits)
    assert e.value.code == -5


@pytest.mark.parametrize("k,ext", [(3, 2), (8, 2), (9, 1), (12, 2), (16, 2)])
def test_extended_to_coeff_roundtrip_and_oracle(api, O, k, ext):
    """extended_to_coeff(coeff_to_extended(c)) == c zero-extended, and bit-exact against the oracle on arbitrary input"""
    rng = np.random.default_rng(90 + k)
    coeffs = O.random_fr(rng, 2 << k).reshape(2, 1 << k, 4)
    e = api.coeff_to_extended(coeffs, ext)
    back = api.extended_to_coeff(e, k, ext)
    assert np.array_equal(back[:, : 1 << k], coeffs) and not back[:, 1 << k:].any()
    arbitrary = O.random_fr(rng, 1 << (k + ext)).reshape(1, 1 << (k + ext), 4)
    assert np.array_equal(api.extended_to_coeff(arbitrary, k, ext), O.extended_to_coeff(arbitrary, k, ext))


def _gate_quotient(api, O, cols, flags, n_cells, bp, k, ext, y):
    """device pipeline: selectors -> polynomials -> extended coset -> gate numerator -> / (X^n - 1) -> coefficients"""
    import ctypes
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    n_cols, n, ne = cols.shape[0], 1 << k, 1 << (k + ext)
    d_flags = api.DeviceBuffer(max(flags.nbytes, 32))
    d_flags.upload(flags)
    d_q = api.DeviceBuffer(n_cols * n * 32)
    check(lib.vdb_layout_selectors_dev(d_flags.ptr, ctypes.c_uint64(n_cells), api._p(bp), ctypes.c_uint64(len(bp)), k, d_q.ptr))
    q = d_q.download((n_cols, n, 4))
    adv_ext = api.coeff_to_extended(api.lagrange_to_coeff(cols), ext)
    sel_ext = api.coeff_to_extended(api.lagrange_to_coeff(q), ext)
    d_a, d_s, d_h = api.DeviceBuffer(adv_ext.nbytes), api.DeviceBuffer(sel_ext.nbytes), api.DeviceBuffer(ne * 32)
    d_a.upload(adv_ext)
    d_s.upload(sel_ext)
    check(lib.vdb_memset_dev(d_h.ptr, 0, ne * 32))
    check(lib.vdb_gate_eval_dev(d_a.ptr, d_s.ptr, ctypes.c_size_t(n_cols), k, ext, api._p(y), d_h.ptr))
    api.sync()
    numer = d_h.download((ne, 4))
    check(lib.vdb_divide_by_vanishing_dev(d_h.ptr, k, ext))
    check(lib.vdb_extended_to_coeff_dev(d_h.ptr, ctypes.c_size_t(1), k, ext))
    api.sync()
    h = d_h.download((ne, 4))
    for b in (d_flags, d_q, d_a, d_s, d_h):
        b.free()
    return q, adv_ext, sel_ext, numer, h


def test_gate_quotient_vanishes_exactly_when_the_gates_hold(api, O):
    """The gate part of the quotient, end to end on a real witness: the numerator sum_c y^c q_c (a + a(w) a(w^2) - a(w^3))
    on the extended coset, divided by X^n - 1, is a polynomial of degree < 2n - 2 exactly when every gate row satisfies
    a + b c = d — a property of the mathematics, independent of any recalled upstream detail.  Also checked: the numerator
    against a numpy restatement on the oracle's field arithmetic, and one broken cell makes the division inexact."""
    rng = np.random.default_rng(2024)
    k, ext = 10, 2
    n, ne = 1 << k, 1 << (k + ext)
    qa, qb = O.quantize(rng.uniform(-3, 3, (3, 6))), O.quantize(rng.uniform(-3, 3, (3, 6)))
    w = api.wit_distance("euclidean", qa, qb, L=9, selectors=True)
    flags, stream = w["flags"], w["stream"]
    bp = api.layout_plan(flags, k)
    cols, _ = api.layout_columns(stream, bp, k)
    cols[:, n - 6:] = O.random_fr(rng, cols.shape[0] * 6).reshape(cols.shape[0], 6, 4)     # blinding rows: no gate there
    y = O.random_fr(rng, 1)[0]
    q, adv_ext, sel_ext, numer, h = _gate_quotient(api, O, cols, flags, len(stream), bp, k, ext, y)
    assert q[:, :, 0].any() and not q[:, n - 9:].any()
    # numpy restatement of the numerator on the oracle's arithmetic
    acc = np.zeros((ne, 4), dtype=np.uint64)
    r = 1 << ext
    for c in range(cols.shape[0]):
        a = adv_ext[c]
        g = O.fr_sub(O.fr_add(a, O.fr_mul(np.roll(a, -r, axis=0), np.roll(a, -2 * r, axis=0))), np.roll(a, -3 * r, axis=0))
        acc = O.fr_add(O.fr_mul(acc, np.tile(y, (ne, 1))), O.fr_mul(sel_ext[c], g))
    assert np.array_equal(numer, acc)
    # exact division: the quotient has degree < 2n - 2
    assert h[: 2 * n - 2].any() and not h[2 * n - 2:].any()
    # break one gate: change the output cell of the first gate of column 0
    row = int(np.flatnonzero(q[0, :, 0] | q[0, :, 1] | q[0, :, 2] | q[0, :, 3])[0])
    bad = cols.copy()
    bad[0, row + 3] = O.fr_add(bad[0, row + 3].reshape(1, 4), O.fr_from_ints([1]))[0]
    _, _, _, _, h_bad = _gate_quotient(api, O, bad, flags, len(stream), bp, k, ext, y)
    assert h_bad[2 * n - 2:].any()


def _copy_cycles(rng, cols_ints, n_cols, n, usable):
    """a copy-constraint permutation over the usable rows: cells of equal value are linked into cycles (as halo2's
    keygen does for every constrain_equal); returns mapping words col' << 32 | row'"""
    mapping = np.array([[(c << 32) | r for r in range(n)] for c in range(n_cols)], dtype=np.uint64)
    groups = {}
    for c in range(n_cols):
        for r in range(usable):
            groups.setdefault(cols_ints[c][r], []).append((c, r))
    n_linked = 0
    for cells in groups.values():
        if len(cells) > 1:
            order = [cells[i] for i in rng.permutation(len(cells))]
            for (c, r), (c2, r2) in zip(order, order[1:] + order[:1]):
                mapping[c, r] = (c2 << 32) | r2
            n_linked += len(cells)
    return mapping, n_linked


@pytest.mark.parametrize("k,n_cols,chunk_len", [(4, 1, 1), (5, 3, 2), (6, 7, 3), (8, 4, 3)])
def test_permutation_sigma_and_product_match_oracle(api, O, k, n_cols, chunk_len):
    rng = np.random.default_rng(500 + k)
    n, usable = 1 << k, (1 << k) - 6
    vals = rng.integers(0, 3 if k < 6 else 9, size=(n_cols, n))   # few distinct values: long copy cycles
    cols = O.fr_from_ints([int(v) for v in vals.reshape(-1)]).reshape(n_cols, n, 4)
    mapping, n_linked = _copy_cycles(rng, vals.tolist(), n_cols, n, usable)
    assert n_linked >= n_cols * usable // 2
    sigma = api.permutation_sigma(mapping, k)
    assert np.array_equal(sigma, O.permutation_sigma(mapping, k))
    beta, gamma = O.random_fr(rng, 1)[0], O.random_fr(rng, 1)[0]
    z = api.permutation_product(cols, sigma, usable, chunk_len, beta, gamma)
    assert np.array_equal(z, O.permutation_product(cols, sigma, usable, chunk_len, beta, gamma))
    one = O.fr_from_ints([1])[0]
    assert np.array_equal(z[0, 0], one) and np.array_equal(z[-1, usable], one) and not z[:, usable + 1:].any()
    # a cell that differs from the cell it is tied to: the product no longer closes
    bad = cols.copy()
    c, r = [(c, r) for c in range(n_cols) for r in range(usable) if int(mapping[c, r]) != (c << 32) | r][0]
    bad[c, r] = O.fr_from_ints([1000])[0]
    zb = api.permutation_product(bad, sigma, usable, chunk_len, beta, gamma)
    assert not np.array_equal(zb[-1, usable], one)
    with pytest.raises(api.VdbError):
        wrong = mapping.copy()
        wrong[0, 0] = np.uint64(n_cols << 32)
        api.permutation_sigma(wrong, k)


def test_permutation_product_closes_on_a_laid_out_witness(api, O):
    """the copy constraints the column layout itself creates — the overlap cell that ends one column is the cell that
    starts the next — at k = 12 on a real distance witness: with exactly those cells tied, the chained
    product over all chunks returns to one"""
    rng = np.random.default_rng(77)
    k = 12
    n, usable = 1 << k, (1 << k) - 6
    qa, qb = O.quantize(rng.uniform(-3, 3, (2, 8))), O.quantize(rng.uniform(-3, 3, (2, 8)))
    w = api.wit_distance("euclidean", qa, qb, L=11, selectors=True)
    bp = api.layout_plan(w["flags"], k)
    cols, _ = api.layout_columns(w["stream"], bp, k)
    n_cols = cols.shape[0]
    assert n_cols >= 3
    mapping = np.array([[(c << 32) | r for r in range(n)] for c in range(n_cols)], dtype=np.uint64)
    for c in range(n_cols - 1):
        last = int(bp[c])                                 # break point = the row of the cell the next column repeats
        assert np.array_equal(cols[c, last], cols[c + 1, 0])
        mapping[c, last], mapping[c + 1, 0] = (c + 1) << 32, (c << 32) | last
    sigma = api.permutation_sigma(mapping, k)
    beta, gamma = O.random_fr(rng, 1)[0], O.random_fr(rng, 1)[0]
    z = api.permutation_product(cols, sigma, usable, 3, beta, gamma)
    one = O.fr_from_ints([1])[0]
    assert np.array_equal(z[-1, usable], one) and not np.array_equal(z[0, usable], one)
    # recurrence spot check through the device field product
    i = 1234
    wi = O.fr_from_ints([pow(O.fr_to_ints(O.root_of_unity(k).reshape(1, 4))[0], i, O.R_MOD)])
    num = den = one.reshape(1, 4)
    dl = O.fr_from_ints([1])
    for c in range(3):
        v = O.fr_add(cols[c, i].reshape(1, 4), gamma.reshape(1, 4))
        num = O.fr_mul(num, O.fr_add(v, O.fr_mul(O.fr_mul(dl, beta.reshape(1, 4)), wi)))
        den = O.fr_mul(den, O.fr_add(v, O.fr_mul(beta.reshape(1, 4), sigma[c, i].reshape(1, 4))))
        dl = O.fr_mul(dl, api.fr_delta().reshape(1, 4))
    assert np.array_equal(O.fr_mul(z[0, i + 1].reshape(1, 4), den), O.fr_mul(z[0, i].reshape(1, 4), num))


@pytest.mark.parametrize("n,usable,bits,n_cols", [(64, 58, 4, 2), (1024, 1018, 8, 3), (65536, 65530, 15, 2)])
def test_lookup_product_closes_and_matches_oracle(api, O, n, usable, bits, n_cols):
    rng = np.random.default_rng(61 * bits + n_cols)
    table = _range_table(O, n, bits)
    ins = np.stack([O.fr_from_ints([int(x) for x in rng.integers(0, 1 << bits, size=n)]) for _ in range(n_cols)])
    pa, ps = api.lookup_permute(ins, table, usable, bits)
    beta, gamma = O.random_fr(rng, 1)[0], O.random_fr(rng, 1)[0]
    z = api.lookup_product(ins, table, pa, ps, usable, beta, gamma)
    one = O.fr_from_ints([1])[0]
    for c in range(n_cols):
        assert np.array_equal(z[c, 0], one) and np.array_equal(z[c, usable], one) and not z[c, usable + 1:].any()
    if n <= 1024:
        assert np.array_equal(z, O.lookup_product(ins, table, pa, ps, usable, beta, gamma))
    else:   # the recurrence, through the device field product
        b, g = np.tile(beta, (usable, 1)), np.tile(gamma, (usable, 1))
        lhs = O.fr_mul(z[0, 1:usable + 1], O.fr_mul(O.fr_add(pa[0, :usable], b), O.fr_add(ps[0, :usable], g)))
        rhs = O.fr_mul(z[0, :usable], O.fr_mul(O.fr_add(ins[0, :usable], b), O.fr_add(table[:usable], g)))
        assert np.array_equal(lhs, rhs)
    # an input value swapped for another one: the permuted pair no longer matches and the product does not close
    bad = ins.copy()
    bad[0, 3] = O.fr_add(bad[0, 3].reshape(1, 4), O.fr_from_ints([1]))[0]
    zb = api.lookup_product(bad, table, pa, ps, usable, beta, gamma)
    assert not np.array_equal(zb[0, usable], one)


def _full_quotient(api, O, k, ext, usable, chunk_len, y, beta, gamma, adv, sel, table, sigma, zp, lk, pa, ps, zl):
    """gates, then the permutation argument over `adv`, then the lookup argument over the columns `lk` (which are part of
    `adv`): numerator on the extended coset / (X^n - 1) -> coefficients of h"""
    import ctypes
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    n, ne = 1 << k, 1 << (k + ext)
    lag = np.zeros((3, n, 4), dtype=np.uint64)
    one = O.fr_from_ints([1])[0]
    lag[0, 0] = one                                       # l0
    lag[1, usable] = one                                  # l_last
    lag[2, :usable] = one                                 # l_active = 1 - (l_last + l_blind)
    ext_of = lambda cols: api.coeff_to_extended(api.lagrange_to_coeff(np.ascontiguousarray(cols)), ext)
    arrays = dict(adv=ext_of(adv), sel=ext_of(sel), table=ext_of(table.reshape(1, n, 4)), sigma=ext_of(sigma), zp=ext_of(zp), lk=ext_of(lk), pa=ext_of(pa),
                  ps=ext_of(ps), zl=ext_of(zl), lag=ext_of(lag))
    d = {}
    for name, a in arrays.items():
        d[name] = api.DeviceBuffer(a.nbytes)
        d[name].upload(a)
    d_h = api.DeviceBuffer(ne * 32)
    check(lib.vdb_memset_dev(d_h.ptr, 0, ne * 32))
    l0, ll, la = (ctypes.c_void_p(d["lag"].ptr.value + i * ne * 32) for i in range(3))
    n_gate = sel.shape[0]
    check(lib.vdb_gate_eval_dev(d["adv"].ptr, d["sel"].ptr, ctypes.c_size_t(n_gate), k, ext, api._p(y), d_h.ptr))
    delta = api.fr_delta()
    check(lib.vdb_permutation_eval_dev(d["adv"].ptr, d["sigma"].ptr, d["zp"].ptr, ctypes.c_size_t(adv.shape[0]), ctypes.c_size_t(chunk_len), k, ext,
                                       ctypes.c_size_t(usable), l0, ll, la, api._p(beta), api._p(gamma), api._p(delta), api._p(y), d_h.ptr))
    check(lib.vdb_lookup_eval_dev(d["lk"].ptr, d["table"].ptr, d["pa"].ptr, d["ps"].ptr, d["zl"].ptr, ctypes.c_size_t(lk.shape[0]), k, ext, l0, ll, la,
                                  api._p(beta), api._p(gamma), api._p(y), d_h.ptr))
    check(lib.vdb_divide_by_vanishing_dev(d_h.ptr, k, ext))
    check(lib.vdb_extended_to_coeff_dev(d_h.ptr, ctypes.c_size_t(1), k, ext))
    api.sync()
    h = d_h.download((ne, 4))
    for b in list(d.values()) + [d_h]:
        b.free()
    return h


def _distance_circuit(api, O, rng, chunk_len, k=10, L=9):
    """a Euclidean-distance witness laid out at 2^k rows with everything the prover rounds derive from it: selectors, sigma
    columns (overlap cells tied), both arguments' product columns, blinding rows"""
    import ctypes
    from halo2_vectordb_amd._lib import check
    n, usable = 1 << k, (1 << k) - 6
    qa, qb = O.quantize(rng.uniform(-3, 3, (3, 6))), O.quantize(rng.uniform(-3, 3, (3, 6)))
    w = api.wit_distance("euclidean", qa, qb, L=L, selectors=True)
    bp = api.layout_plan(w["flags"], k)
    cols, lcols = api.layout_columns(w["stream"], bp, k, lookup=w["lookup"])
    n_adv, n_lk = cols.shape[0], lcols.shape[0]
    assert n_adv >= 2 and n_lk >= 1
    lib = api.init()
    d_flags, d_q = api.DeviceBuffer(max(w["flags"].nbytes, 32)), api.DeviceBuffer(n_adv * n * 32)
    d_flags.upload(w["flags"])
    check(lib.vdb_layout_selectors_dev(d_flags.ptr, ctypes.c_uint64(len(w["stream"])), api._p(bp), ctypes.c_uint64(len(bp)), k, d_q.ptr))
    sel = d_q.download((n_adv, n, 4))
    d_flags.free(), d_q.free()
    blind = lambda shape_cols, rows: O.random_fr(rng, shape_cols * rows).reshape(shape_cols, rows, 4)
    allc = np.concatenate([cols, lcols])                  # the permutation covers the gate columns and the lookup columns
    allc[:, usable + 1:] = blind(allc.shape[0], n - usable - 1)
    table = _range_table(O, n, L)
    # copy constraints: the overlap cells between consecutive gate columns
    mapping = np.array([[(c << 32) | r for r in range(n)] for c in range(allc.shape[0])], dtype=np.uint64)
    for c in range(n_adv - 1):
        last = int(bp[c])
        mapping[c, last], mapping[c + 1, 0] = (c + 1) << 32, (c << 32) | last
    sigma = api.permutation_sigma(mapping, k)
    beta, gamma, y = O.random_fr(rng, 1)[0], O.random_fr(rng, 1)[0], O.random_fr(rng, 1)[0]
    zp = api.permutation_product(allc, sigma, usable, chunk_len, beta, gamma)
    zp[:, usable + 1:] = blind(zp.shape[0], n - usable - 1)
    lk = allc[n_adv:]
    pa, ps = api.lookup_permute(lk, table, usable, L)
    zl = api.lookup_product(lk, table, pa, ps, usable, beta, gamma)
    pa[:, usable:], ps[:, usable:] = blind(n_lk, n - usable), blind(n_lk, n - usable)
    zl[:, usable + 1:] = blind(n_lk, n - usable - 1)
    args = dict(adv=allc, sel=sel, table=table, sigma=sigma, zp=zp, lk=lk, pa=pa, ps=ps, zl=zl)
    return args, (beta, gamma, y), usable, n_adv


@pytest.mark.parametrize("chunk_len", [2, 3])
def test_full_quotient_of_a_laid_out_distance_circuit(api, O, chunk_len):
    """Gates + permutation argument + lookup argument together, on a real Euclidean-distance witness laid out at k = 10 with
    its range-check cells in lookup columns: the whole numerator of h(X), built by the device bricks (layout, selectors,
    lookup permutation, both running products, the three evaluation kernels on the extended coset), is divisible by X^n - 1
    — deg h <= (chunk_len + 2)(n - 1) - n — and stops being so when a product column or a tied cell is tampered with.
    The statement is mathematical (the PLONK identities), independent of recalled upstream detail."""
    rng = np.random.default_rng(4242)
    k, ext = 10, 2
    n = 1 << k
    args, (beta, gamma, y), usable, n_adv = _distance_circuit(api, O, rng, chunk_len, k)
    allc, sigma, zl = args["adv"], args["sigma"], args["zl"]
    run = lambda **kw: _full_quotient(api, O, k, ext, usable, chunk_len, y, beta, gamma, **{**args, **kw})
    bound = (chunk_len + 2) * (n - 1) - n + 1             # number of coefficients h may have
    h = run()
    assert h[:bound].any() and not h[bound:].any()
    # tamper with one value of the lookup product inside the usable rows
    zl_bad = zl.copy()
    zl_bad[0, 100] = O.fr_add(zl_bad[0, 100].reshape(1, 4), O.fr_from_ints([1]))[0]
    assert run(zl=zl_bad)[bound:].any()
    # break a tie: the first cell of the second gate column no longer equals the overlap cell it repeats (its gate is also
    # broken, so restrict the claim to: not divisible)
    adv_bad = allc.copy()
    adv_bad[1, 0] = O.fr_add(adv_bad[1, 0].reshape(1, 4), O.fr_from_ints([1]))[0]
    assert run(adv=adv_bad)[bound:].any()
    # a permutation product computed for other challenges
    zp_other = api.permutation_product(allc, sigma, usable, chunk_len, gamma, beta)
    assert run(zp=zp_other)[bound:].any()


def test_verifier_identity_at_a_random_point(api, O):
    """What a verifier checks: every committed polynomial evaluated at x (and at the rotated points) by vdb_eval_polys_dev,
    the gate / permutation / lookup expressions recombined from those evaluations with Python integers, against
    h(x) (x^n - 1) with h from the device quotient pipeline."""
    rng = np.random.default_rng(777)
    k, ext, chunk_len = 10, 2, 3
    n = 1 << k
    R = O.R_MOD
    args, (beta, gamma, y), usable, n_adv = _distance_circuit(api, O, rng, chunk_len, k)
    h = _full_quotient(api, O, k, ext, usable, chunk_len, y, beta, gamma, **args)
    to_int = lambda a: O.fr_to_ints(np.asarray(a).reshape(-1, 4))
    b, g, yv = to_int(beta)[0], to_int(gamma)[0], to_int(y)[0]
    w = to_int(O.root_of_unity(k))[0]
    x = to_int(O.random_fr(rng, 1))[0]
    delta = to_int(api.fr_delta())[0]
    lag = np.zeros((3, n, 4), dtype=np.uint64)
    one = O.fr_from_ints([1])[0]
    lag[0, 0], lag[1, usable], lag[2, :usable] = one, one, one
    coeff = {name: api.lagrange_to_coeff(np.ascontiguousarray(a if a.ndim == 3 else a.reshape(1, n, 4))) for name, a in {**args, "lag": lag}.items()}
    def ev(name, rot=0):
        pt = x * pow(w, rot % n, R) % R
        return to_int(api.eval_polys(coeff[name], O.fr_from_ints([pt])[0]))
    acc = 0
    a0, a1, a2, a3, q = ev("adv"), ev("adv", 1), ev("adv", 2), ev("adv", 3), ev("sel")
    for c in range(n_adv):
        acc = (acc * yv + q[c] * (a0[c] + a1[c] * a2[c] - a3[c])) % R
    l0, ll, la = ev("lag")
    sg, z0, z1, zb = ev("sigma"), ev("zp"), ev("zp", 1), ev("zp", -(n - usable))
    n_cols, n_sets = len(a0), len(z0)
    acc = (acc * yv + l0 * (1 - z0[0])) % R
    acc = (acc * yv + ll * (z0[-1] * z0[-1] - z0[-1])) % R
    for i in range(1, n_sets):
        acc = (acc * yv + l0 * (z0[i] - zb[i - 1])) % R
    cur = b * x % R
    for i in range(n_sets):
        left, right = z1[i], z0[i]
        for c in range(i * chunk_len, min((i + 1) * chunk_len, n_cols)):
            left = left * (a0[c] + b * sg[c] + g) % R
            right = right * (a0[c] + cur + g) % R
            cur = cur * delta % R
        acc = (acc * yv + la * (left - right)) % R
    A, S, PA, PS, PAm, Z, Z1 = ev("lk"), ev("table")[0], ev("pa"), ev("ps"), ev("pa", -1), ev("zl"), ev("zl", 1)
    for c in range(len(A)):
        acc = (acc * yv + l0 * (1 - Z[c])) % R
        acc = (acc * yv + ll * (Z[c] * Z[c] - Z[c])) % R
        acc = (acc * yv + la * (Z1[c] * (PA[c] + b) * (PS[c] + g) - Z[c] * (A[c] + b) * (S + g))) % R
        acc = (acc * yv + l0 * (PA[c] - PS[c])) % R
        acc = (acc * yv + la * (PA[c] - PS[c]) * (PA[c] - PAm[c])) % R
    hx = to_int(api.eval_polys(h.reshape(1, -1, 4), O.fr_from_ints([x])[0]))[0]
    assert acc == hx * (pow(x, n, R) - 1) % R


def _kate_div_ints(a, x, R):
    """halo2 arithmetic::kate_division restated: q_{n-2} = a_{n-1}, q_{i-1} = a_i + x q_i"""
    q = [0] * len(a)
    s = 0
    for i in range(len(a) - 1, 0, -1):
        s = (a[i] + x * s) % R
        q[i - 1] = s
    return q, (a[0] + x * s) % R


@pytest.mark.parametrize("n_cols,n", [(1, 1), (2, 2), (3, 7), (2, 2048), (2, 2049), (3, 4100), (2, 65536)])
def test_kate_division_matches_restatement(api, O, n_cols, n):
    rng = np.random.default_rng(300 + n)
    coeffs = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    for x in (O.random_fr(rng, 1)[0], O.fr_from_ints([0])[0], O.fr_from_ints([1])[0]):
        q, rem = api.kate_div(coeffs, x)
        xi = O.fr_to_ints(x.reshape(1, 4))[0]
        assert np.array_equal(rem, api.eval_polys(coeffs, x))
        for c in range(n_cols if n <= 4100 else 1):
            want_q, want_r = _kate_div_ints(O.fr_to_ints(coeffs[c]), xi, O.R_MOD)
            assert O.fr_to_ints(q[c]) == want_q and O.fr_to_ints(rem[c].reshape(1, 4))[0] == want_r


def test_poly_lincomb_is_horner_over_the_polynomials(api, O):
    rng = np.random.default_rng(9)
    n_cols, n = 5, 3000
    polys = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    v = O.random_fr(rng, 1)[0]
    acc = np.zeros((n, 4), dtype=np.uint64)
    for c in range(n_cols):
        acc = O.fr_add(O.fr_mul(acc, np.tile(v, (n, 1))), polys[c])
    assert np.array_equal(api.poly_lincomb(polys, v), acc)


def test_kzg_opening_verifies_in_the_exponent(api, O):
    """a complete batched KZG opening from the bricks: commit the polynomials, combine them with powers of v, divide by
    (X - x), commit the quotient; the pairing check e(C - [y]G, H) = e(W, [tau - x]H) is verified in the exponent, which the
    test can do because it knows tau:  C - [y]G == [tau - x] W  as group elements (oracle group arithmetic)."""
    rng = np.random.default_rng(1234)
    k, n_polys = 10, 4
    n = 1 << k
    tau = 0xC0FFEE1234567
    g, _ = O.srs_from_tau(k, tau)
    srs = api.Srs(k, g, None)
    polys = O.random_fr(rng, n_polys * n).reshape(n_polys, n, 4)
    commits = api.msm_batch(srs, polys, basis=0)
    v, x = O.random_fr(rng, 1)[0], O.random_fr(rng, 1)[0]
    comb = api.poly_lincomb(polys, v)
    q, rem = api.kate_div(comb.reshape(1, n, 4), x)
    W = api.msm(srs, q[0], basis=0)
    srs.free()
    R = O.R_MOD
    vi, xi, yi = (O.fr_to_ints(a.reshape(1, 4))[0] for a in (v, x, rem[0]))
    # the combined evaluation is the same combination of the individual evaluations
    evals = O.fr_to_ints(api.eval_polys(polys, x))
    acc = 0
    for e in evals:
        acc = (acc * vi + e) % R
    assert acc == yi
    # C = sum v^(m-1-c) C_c (the verifier's side), minus [y] G
    powers = [pow(vi, n_polys - 1 - c, R) for c in range(n_polys)] + [(-yi) % R]
    lhs = O.msm_naive(O.fr_from_ints(powers), np.concatenate([commits, O.g1_generator().reshape(1, 8)]))
    rhs = O.msm_naive(O.fr_from_ints([(tau - xi) % R]), W.reshape(1, 8))
    assert np.array_equal(lhs, rhs) and lhs.any()


def test_prover_round_entry_points_reject_bad_arguments_and_accept_empty_batches(api, O):
    import ctypes
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    rng = np.random.default_rng(3)
    one_fr = O.fr_from_ints([1])[0]
    buf = api.DeviceBuffer(64 * 32)
    buf.upload(O.random_fr(rng, 64))
    sz = ctypes.c_size_t
    # empty batches are no-ops
    check(lib.vdb_permutation_product_dev(buf.ptr, buf.ptr, sz(0), 4, sz(10), sz(3), api._p(one_fr), api._p(one_fr), api._p(one_fr), buf.ptr))
    check(lib.vdb_lookup_product_dev(buf.ptr, buf.ptr, buf.ptr, buf.ptr, sz(0), sz(16), sz(10), api._p(one_fr), api._p(one_fr), buf.ptr))
    check(lib.vdb_kate_div_dev(buf.ptr, sz(0), sz(16), api._p(one_fr), buf.ptr, None))
    check(lib.vdb_poly_lincomb_dev(buf.ptr, sz(0), sz(16), api._p(one_fr), buf.ptr))
    check(lib.vdb_fill_rows_dev(buf.ptr, sz(2), sz(16), sz(16), buf.ptr))
    # bad arguments come back as VDB_ERR_ARG, never as a fault
    for call in (lambda: lib.vdb_permutation_product_dev(buf.ptr, buf.ptr, sz(2), 4, sz(16), sz(3), api._p(one_fr), api._p(one_fr), api._p(one_fr), buf.ptr),   # usable_rows == n
                 lambda: lib.vdb_permutation_product_dev(buf.ptr, buf.ptr, sz(2), 4, sz(10), sz(0), api._p(one_fr), api._p(one_fr), api._p(one_fr), buf.ptr),   # chunk_len 0
                 lambda: lib.vdb_lookup_product_dev(buf.ptr, buf.ptr, buf.ptr, buf.ptr, sz(1), sz(16), sz(16), api._p(one_fr), api._p(one_fr), buf.ptr),
                 lambda: lib.vdb_kate_div_dev(None, sz(1), sz(16), api._p(one_fr), buf.ptr, None),
                 lambda: lib.vdb_fill_rows_dev(buf.ptr, sz(1), sz(16), sz(17), buf.ptr),
                 lambda: lib.vdb_permutation_eval_range_dev(buf.ptr, buf.ptr, buf.ptr, sz(6), sz(3), 2, 2, sz(2), buf.ptr, buf.ptr, buf.ptr, api._p(one_fr), api._p(one_fr),
                                                            api._p(one_fr), api._p(one_fr), buf.ptr, sz(1), sz(3))):   # set range beyond the 2 sets
        assert call() == -3           # VDB_ERR_ARG
    # usable_rows = 0: the products are the constant one at row 0 and zero elsewhere
    cols = O.random_fr(rng, 2 * 16).reshape(2, 16, 4)
    z = api.permutation_product(cols, cols, 0, 3, one_fr, one_fr)
    assert np.array_equal(z[0, 0], one_fr) and not z[:, 1:].any()
    # a chunk longer than the column count is one set
    z = api.permutation_product(cols, cols, 10, 5, one_fr, one_fr)
    assert z.shape[0] == 1
    buf.free()


def test_gate_eval_on_the_sub_coset(api, O):
    """vdb_gate_eval_sub_dev: the gates evaluated on the coset of 2 n points that lies inside the 4 n points the advice cosets were
    made for, against selector cosets made for 2 n points — the same accumulator, bit for bit, as vdb_gate_eval_dev on cosets
    that were all made for 2 n points; and divided by X^n - 1 there, the quotient's coefficients equal the ones the 4 n route
    gives (the gate has degree 3: its quotient share has degree below 2 n) on a witness whose gates hold."""
    import ctypes
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    rng = np.random.default_rng(77)
    k = 9
    n = 1 << k
    qa, qb = O.quantize(rng.uniform(-3, 3, (2, 5))), O.quantize(rng.uniform(-3, 3, (2, 5)))
    w = api.wit_distance("manhattan", qa, qb, L=8, selectors=True)
    bp = api.layout_plan(w["flags"], k)
    cols, _ = api.layout_columns(w["stream"], bp, k)
    n_cols = cols.shape[0]
    y = O.random_fr(rng, 1)[0]
    q, _, _, _, h4 = _gate_quotient(api, O, cols, w["flags"], len(w["stream"]), bp, k, 2, y)
    adv_c, sel_c = api.lagrange_to_coeff(cols), api.lagrange_to_coeff(q)
    adv4, adv2, sel2 = api.coeff_to_extended(adv_c, 2), api.coeff_to_extended(adv_c, 1), api.coeff_to_extended(sel_c, 1)
    assert np.array_equal(adv4[:, ::2], adv2)            # the small coset is every second point of the large one
    bufs = [api.DeviceBuffer(a.nbytes) for a in (adv4, adv2, sel2)]
    for b, a in zip(bufs, (adv4, adv2, sel2)):
        b.upload(a)
    d_h1, d_h2 = api.DeviceBuffer(2 * n * 32), api.DeviceBuffer(2 * n * 32)
    for d in (d_h1, d_h2):
        check(lib.vdb_memset_dev(d.ptr, 0, 2 * n * 32))
    check(lib.vdb_gate_eval_sub_dev(bufs[0].ptr, 2, bufs[2].ptr, ctypes.c_size_t(n_cols), k, 1, api._p(y), d_h1.ptr))
    check(lib.vdb_gate_eval_dev(bufs[1].ptr, bufs[2].ptr, ctypes.c_size_t(n_cols), k, 1, api._p(y), d_h2.ptr))
    api.sync()
    assert np.array_equal(d_h1.download((2 * n, 4)), d_h2.download((2 * n, 4)))
    check(lib.vdb_divide_by_vanishing_dev(d_h1.ptr, k, 1))
    check(lib.vdb_extended_to_coeff_dev(d_h1.ptr, ctypes.c_size_t(1), k, 1))
    api.sync()
    h2 = d_h1.download((2 * n, 4))
    assert np.array_equal(h2, h4[: 2 * n]) and not h4[2 * n:].any()
    assert lib.vdb_gate_eval_sub_dev(bufs[0].ptr, 0, bufs[2].ptr, ctypes.c_size_t(n_cols), k, 1, api._p(y), d_h1.ptr) == -3     # VDB_ERR_ARG: adv_ext_k < ext_k
    before = d_h2.download((2 * n, 4))
    check(lib.vdb_gate_eval_sub_dev(bufs[0].ptr, 2, bufs[2].ptr, ctypes.c_size_t(0), k, 1, api._p(y), d_h2.ptr))                  # no columns: nothing happens
    api.sync()
    assert np.array_equal(d_h2.download((2 * n, 4)), before)
    for b in bufs + [d_h1, d_h2]:
        b.free()


def test_scaled_coset_extension_and_prescaled_sigma(api, O):
    """vdb_coeff_to_extended_scaled_dev(c, s) is the coset image of s c(X) — bit for bit vdb_coeff_to_extended_dev of the scaled
    coefficients —, and the permutation part of the quotient fed with the cosets of beta sigma (head bit 1) is, bit for bit, the
    accumulator it computes from the cosets of sigma with its own product by beta."""
    import ctypes
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    sz = ctypes.c_size_t
    rng = np.random.default_rng(4242)
    k, ext, chunk = 9, 2, 3
    n, ne = 1 << k, 1 << (k + ext)
    n_cols = 7
    coeff = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    s = O.random_fr(rng, 1)[0]
    d_c, d_e1, d_e2, d_sc = (api.DeviceBuffer(n_cols * m * 32) for m in (n, ne, ne, n))
    d_c.upload(coeff)
    check(lib.vdb_coeff_to_extended_scaled_dev(d_c.ptr, d_e1.ptr, sz(n_cols), k, ext, api._p(s)))
    scaled = O.fr_mul(coeff.reshape(-1, 4), np.broadcast_to(s, (n_cols * n, 4)).copy()).reshape(n_cols, n, 4)
    d_sc.upload(scaled)
    check(lib.vdb_coeff_to_extended_dev(d_sc.ptr, d_e2.ptr, sz(n_cols), k, ext))
    got, want = d_e1.download((n_cols, ne, 4)), d_e2.download((n_cols, ne, 4))
    assert np.array_equal(got, want)
    _, want_o = O.lde_batch(api.ntt_batch(scaled[:2], api.root_of_unity(k)), ext=ext, threads=2)     # (lde_batch takes Lagrange form)
    assert np.array_equal(got[:2], want_o)
    # the permutation terms: adv / sigma / z cosets of random polynomials, two sets + a short third
    usable = n - 6
    beta, gamma, y = O.random_fr(rng, 3)
    adv_c, sig_c, z_c = (O.random_fr(rng, m * n).reshape(m, n, 4) for m in (n_cols, n_cols, 3))
    lag_c = O.random_fr(rng, 3 * n).reshape(3, n, 4)
    bufs = {}
    for name, c in (("adv", adv_c), ("sig", sig_c), ("z", z_c), ("lag", lag_c)):
        bc, be = api.DeviceBuffer(c.shape[0] * n * 32), api.DeviceBuffer(c.shape[0] * ne * 32)
        bc.upload(c)
        check(lib.vdb_coeff_to_extended_dev(bc.ptr, be.ptr, sz(c.shape[0]), k, ext))
        bufs[name] = (bc, be)
    d_sig_b = api.DeviceBuffer(n_cols * ne * 32)
    check(lib.vdb_coeff_to_extended_scaled_dev(bufs["sig"][0].ptr, d_sig_b.ptr, sz(n_cols), k, ext, api._p(beta)))
    l0, ll, la = (ctypes.c_void_p(bufs["lag"][1].ptr.value + i * ne * 32) for i in range(3))
    acc0 = O.random_fr(rng, ne)
    outs = []
    for sig_ptr, head in ((bufs["sig"][1].ptr, 0), (d_sig_b.ptr, 2)):
        d_acc = api.DeviceBuffer(ne * 32)
        d_acc.upload(acc0)
        check(lib.vdb_permutation_eval_parts_dev(bufs["adv"][1].ptr, sz(0), sig_ptr, bufs["z"][1].ptr, sz(0), None, None, sz(n_cols), sz(chunk), k, ext, sz(usable),
                                                 l0, ll, la, api._p(beta), api._p(gamma), api._p(api.fr_delta()), api._p(y), d_acc.ptr, head, sz(0), sz(0), sz(0), sz(3)))
        outs.append(d_acc.download((ne, 4)))
        d_acc.free()
    assert np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], acc0)
    for bc, be in bufs.values():
        bc.free()
        be.free()
    for b in (d_c, d_e1, d_e2, d_sc, d_sig_b):
        b.free()


REV2 = (0, 2, 1, 3)          # slot t of the extended domain taken coset by coset = points 4 r + REV2[t] of the natural order


def _to_slots(ext, n_slots):
    """(cols, 4 n, 4) in natural order -> (cols, n_slots, n, 4) coset by coset"""
    return np.stack([ext[:, REV2[t]::4] for t in range(n_slots)], axis=1)


@pytest.mark.parametrize("k", [3, 9, 10, 11, 12, 16, 18])
def test_cosets_of_the_extended_domain_one_by_one(api, O, k):
    """vdb_coeff_to_cosets_dev: slot t, row r == point 4 r + bitrev2(t) of vdb_coeff_to_extended_dev, bit for bit, for 1..4 slots, with
    and without a scalar (single-pass and multi-pass sizes)"""
    import ctypes
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    sz = ctypes.c_size_t
    rng = np.random.default_rng(100 + k)
    n, ne = 1 << k, 4 << k
    n_cols = 5 if k <= 12 else (3 if k <= 16 else 2)     # (k = 18: the 2^20-point reference transform runs in three passes)
    coeff = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    s = O.random_fr(rng, 1)[0]
    d_c, d_e, d_s = api.DeviceBuffer(coeff.nbytes), api.DeviceBuffer(n_cols * ne * 32), api.DeviceBuffer(n_cols * ne * 32)
    d_c.upload(coeff)
    try:
        for scale in (None, s):
            if scale is None:
                check(lib.vdb_coeff_to_extended_dev(d_c.ptr, d_e.ptr, sz(n_cols), k, 2))
            else:
                check(lib.vdb_coeff_to_extended_scaled_dev(d_c.ptr, d_e.ptr, sz(n_cols), k, 2, api._p(scale)))
            ext = d_e.download((n_cols, ne, 4))
            for n_slots in (1, 2, 3, 4):
                check(lib.vdb_memset_dev(d_s.ptr, 0xEE, sz(n_cols * ne * 32)))
                check(lib.vdb_coeff_to_cosets_dev(d_c.ptr, d_s.ptr, sz(n_cols), k, n_slots, None if scale is None else api._p(scale)))
                got = d_s.download((n_cols, n_slots, n, 4))
                assert np.array_equal(got, _to_slots(ext, n_slots)), (k, n_slots, scale is not None)
        assert lib.vdb_coeff_to_cosets_dev(d_c.ptr, d_s.ptr, sz(n_cols), k, 5, None) == -3
        check(lib.vdb_coeff_to_cosets_dev(d_c.ptr, d_s.ptr, sz(0), k, 3, None))
    finally:
        for b in (d_c, d_e, d_s):
            b.free()


@pytest.mark.parametrize("k,n_slots", [(4, 3), (9, 2), (9, 3), (11, 3), (12, 4), (16, 3)])
def test_quotient_back_from_its_cosets(api, O, k, n_slots):
    """vdb_cosets_to_coeff_dev: for a polynomial h of degree below n_slots n, the values of h (X^n - 1) on the first n_slots cosets give
    back h's pieces exactly — and they are what the 4 n route (divide_by_vanishing + extended_to_coeff) gives"""
    import ctypes
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    sz = ctypes.c_size_t
    rng = np.random.default_rng(7 * k + n_slots)
    n, ne = 1 << k, 4 << k
    h = np.zeros((ne, 4), dtype=np.uint64)
    h[: n_slots * n] = O.random_fr(rng, n_slots * n)
    # the numerator h (X^n - 1) on the extended domain: coeff_to_extended of each piece, joined with (X^n)^j and multiplied point by point
    pieces = h[: n_slots * n].reshape(n_slots, n, 4)
    d_p, d_e = api.DeviceBuffer(pieces.nbytes), api.DeviceBuffer(n_slots * ne * 32)
    d_p.upload(pieces)
    check(lib.vdb_coeff_to_extended_dev(d_p.ptr, d_e.ptr, sz(n_slots), k, 2))
    pe = d_e.download((n_slots, ne, 4))
    R = O.R_MOD
    zeta = O.fr_to_ints(np.asarray(O.zeta()).reshape(1, 4))[0]
    w4n = O.fr_to_ints(np.asarray(api.root_of_unity(k + 2)).reshape(1, 4))[0]
    vals = [O.fr_to_ints(pe[j]) for j in range(n_slots)]
    xns = [pow(zeta * pow(w4n, c, R) % R, n, R) for c in range(4)]          # X^n depends on the point's coset only
    out = []
    for j4 in range(ne):
        xn = xns[j4 & 3]
        hv = sum(vals[j][j4] * pow(xn, j, R) for j in range(n_slots)) % R
        out.append(hv * (xn - 1) % R)
    num = O.fr_from_ints(out)
    slots = np.ascontiguousarray(np.stack([num[REV2[t]::4] for t in range(n_slots)]))
    d_s, d_out, d_n = api.DeviceBuffer(slots.nbytes), api.DeviceBuffer(n_slots * n * 32), api.DeviceBuffer(ne * 32)
    d_s.upload(slots)
    check(lib.vdb_cosets_to_coeff_dev(d_s.ptr, d_out.ptr, k, n_slots))
    got = d_out.download((n_slots, n, 4))
    assert np.array_equal(got, pieces)
    d_n.upload(num)
    check(lib.vdb_divide_by_vanishing_dev(d_n.ptr, k, 2))
    check(lib.vdb_extended_to_coeff_dev(d_n.ptr, sz(1), k, 2))
    assert np.array_equal(d_n.download((ne, 4)), h)
    for b in (d_p, d_e, d_s, d_out, d_n):
        b.free()


def test_quotient_terms_coset_by_coset_are_the_natural_order_ones(api, O):
    """the permutation, lookup and gate parts of the numerator on arrays laid out coset by coset (three slots; the gates two) against
    the same kernels on the natural order of the 4 n points: equal accumulators, point for point"""
    import ctypes
    from halo2_vectordb_amd._lib import check
    lib = api.init()
    sz = ctypes.c_size_t
    rng = np.random.default_rng(90210)
    k, chunk = 9, 2
    n, ne = 1 << k, 4 << k
    n_cols, n_sets, n_lk = 7, 4, 2
    usable = n - 7
    beta, gamma, y = O.random_fr(rng, 3)

    def make(m):
        c = O.random_fr(rng, m * n).reshape(m, n, 4)
        bc, be, bs = api.DeviceBuffer(c.nbytes), api.DeviceBuffer(m * ne * 32), api.DeviceBuffer(m * 3 * n * 32)
        bc.upload(c)
        check(lib.vdb_coeff_to_extended_dev(bc.ptr, be.ptr, sz(m), k, 2))
        check(lib.vdb_coeff_to_cosets_dev(bc.ptr, bs.ptr, sz(m), k, 3, None))
        return bc, be, bs
    B = {name: make(m) for name, m in (("adv", n_cols), ("sig", n_cols), ("z", n_sets), ("lag", 3), ("tab", 1), ("pa", n_lk), ("ps", n_lk), ("zl", n_lk), ("sel", n_cols))}
    acc0 = O.random_fr(rng, ne)
    slots0 = np.ascontiguousarray(np.stack([acc0[REV2[t]::4] for t in range(3)]))

    def lag(which, i):
        buf = B["lag"][which]
        return ctypes.c_void_p(buf.ptr.value + i * (ne if which == 1 else 3 * n) * 32)
    d_a, d_s = api.DeviceBuffer(ne * 32), api.DeviceBuffer(3 * n * 32)
    try:
        # permutation: head terms, chaining, product terms in one call each
        d_a.upload(acc0)
        d_s.upload(slots0)
        zl_ptr = lambda which, i: ctypes.c_void_p(B["z"][which].ptr.value + i * (ne if which == 1 else 3 * n) * 32)
        check(lib.vdb_permutation_eval_parts_dev(B["adv"][1].ptr, sz(0), B["sig"][1].ptr, B["z"][1].ptr, sz(0), zl_ptr(1, 0), zl_ptr(1, n_sets - 1), sz(n_cols), sz(chunk), k, 2,
                                                 sz(usable), lag(1, 0), lag(1, 1), lag(1, 2), api._p(beta), api._p(gamma), api._p(api.fr_delta()), api._p(y), d_a.ptr, 1,
                                                 sz(1), sz(n_sets), sz(0), sz(n_sets)))
        check(lib.vdb_permutation_eval_parts_cosets_dev(B["adv"][2].ptr, sz(0), B["sig"][2].ptr, B["z"][2].ptr, sz(0), zl_ptr(2, 0), zl_ptr(2, n_sets - 1), sz(n_cols), sz(chunk),
                                                        k, 3, sz(usable), lag(2, 0), lag(2, 1), lag(2, 2), api._p(beta), api._p(gamma), api._p(api.fr_delta()), api._p(y),
                                                        d_s.ptr, 1, sz(1), sz(n_sets), sz(0), sz(n_sets)))
        nat, got = d_a.download((ne, 4)), d_s.download((3, n, 4))
        assert not np.array_equal(nat, acc0)
        for t in range(3):
            assert np.array_equal(got[t], nat[REV2[t]::4]), t
        # lookup argument
        check(lib.vdb_lookup_eval_dev(B["adv"][1].ptr, B["tab"][1].ptr, B["pa"][1].ptr, B["ps"][1].ptr, B["zl"][1].ptr, sz(n_lk), k, 2, lag(1, 0), lag(1, 1), lag(1, 2),
                                      api._p(beta), api._p(gamma), api._p(y), d_a.ptr))
        check(lib.vdb_lookup_eval_cosets_dev(B["adv"][2].ptr, B["tab"][2].ptr, B["pa"][2].ptr, B["ps"][2].ptr, B["zl"][2].ptr, sz(n_lk), k, 3, lag(2, 0), lag(2, 1), lag(2, 2),
                                             api._p(beta), api._p(gamma), api._p(y), d_s.ptr))
        nat2, got2 = d_a.download((ne, 4)), d_s.download((3, n, 4))
        assert not np.array_equal(nat2, nat)
        for t in range(3):
            assert np.array_equal(got2[t], nat2[REV2[t]::4]), t
        # gates: two slots = the coset of 2 n points; selectors made for two slots only
        d_sel2, d_g, d_gs = api.DeviceBuffer(n_cols * 2 * n * 32), api.DeviceBuffer(2 * n * 32), api.DeviceBuffer(2 * n * 32)
        d_sel_nat = api.DeviceBuffer(n_cols * 2 * n * 32)
        check(lib.vdb_coeff_to_cosets_dev(B["sel"][0].ptr, d_sel2.ptr, sz(n_cols), k, 2, None))
        check(lib.vdb_coeff_to_extended_dev(B["sel"][0].ptr, d_sel_nat.ptr, sz(n_cols), k, 1))
        g0 = O.random_fr(rng, 2 * n)
        d_g.upload(g0)
        d_gs.upload(np.ascontiguousarray(np.stack([g0[0::2], g0[1::2]])))
        check(lib.vdb_gate_eval_sub_dev(B["adv"][1].ptr, 2, d_sel_nat.ptr, sz(n_cols), k, 1, api._p(y), d_g.ptr))
        check(lib.vdb_gate_eval_cosets_dev(B["adv"][2].ptr, 3, d_sel2.ptr, sz(n_cols), k, 2, api._p(y), d_gs.ptr))
        natg, gotg = d_g.download((2 * n, 4)), d_gs.download((2, n, 4))
        assert np.array_equal(gotg[0], natg[0::2]) and np.array_equal(gotg[1], natg[1::2]) and not np.array_equal(natg, g0)
        for b in (d_sel2, d_g, d_gs, d_sel_nat):
            b.free()
    finally:
        for bufs in B.values():
            for b in bufs:
                b.free()
        d_a.free()
        d_s.free()


def test_lookup_permute_sweep_of_shapes_and_value_patterns(api, O):
    """the lookup argument's permuted columns on inputs no witness produces: every column height 2^6 .. 2^12 with table widths from 2 bits
    up, one value only, two values, every table entry exactly once (as far as the rows reach), the largest entry alone, values that
    appear once among long runs — against the oracle's construction, with the two properties the argument needs"""
    rng = np.random.default_rng(8086)
    for case in range(18):
        k = int(rng.integers(6, 13))
        n = 1 << k
        usable = n - 6
        bits = int(rng.integers(2, min(k, 12)))
        table = _range_table(O, n, bits)
        tab_ints = O.fr_to_ints(table[:usable])
        m = 1 << bits
        cols = []
        for pattern in range(int(rng.integers(1, 5))):
            kind = (case + pattern) % 6
            if kind == 0:
                v = np.full(n, int(rng.integers(0, m)))
            elif kind == 1:
                v = rng.choice([0, m - 1], size=n)
            elif kind == 2:
                v = np.arange(n) % m
            elif kind == 3:
                v = np.full(n, m - 1)
            elif kind == 4:
                v = np.zeros(n, dtype=np.int64)
                v[rng.integers(0, usable, 5)] = rng.integers(0, m, 5)
            else:
                v = rng.integers(0, m, size=n)
            cols.append(O.fr_from_ints([int(x) for x in v]))
        ins = np.stack(cols)
        got_a, got_s = api.lookup_permute(ins, table, usable, bits)
        for c in range(len(cols)):
            want_a, want_s = O.lookup_permute(O.fr_to_ints(ins[c, :usable]), tab_ints)
            assert O.fr_to_ints(got_a[c, :usable]) == want_a, (case, c, k, bits)
            assert O.fr_to_ints(got_s[c, :usable]) == want_s, (case, c, k, bits)
            assert not got_a[c, usable:].any() and not got_s[c, usable:].any()
            assert sorted(want_s) == sorted(tab_ints)
            assert all(want_a[i] == want_s[i] or (i and want_a[i] == want_a[i - 1]) for i in range(usable))


def test_polynomial_bricks_sweep_over_odd_lengths(api, O):
    """grand products, evaluation, division by (X - x) and the Horner combination at lengths and batch widths that are no power of two
    and no multiple of a workgroup (1 .. 5000 coefficients, 1 .. 9 polynomials), at random points and at 0, 1, -1 and a root of unity —
    against the oracle / the Python-integer restatement"""
    rng = np.random.default_rng(24601)
    R = O.R_MOD
    for case in range(16):
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 257, 1000, 1023, 1025, 2049, 4097])) if case % 2 else int(rng.integers(1, 5000))
        n_cols = int(rng.integers(1, 10))
        polys = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
        pts = [O.random_fr(rng, 1)[0], O.fr_from_ints([0])[0], O.fr_from_ints([1])[0], O.fr_from_ints([R - 1])[0], O.root_of_unity(int(rng.integers(1, 12)))]
        x = pts[case % len(pts)]
        xi = O.fr_to_ints(x.reshape(1, 4))[0]
        ev = api.eval_polys(polys, x)
        assert np.array_equal(ev, O.eval_polys(polys, x)), (case, n, n_cols)
        q, rem = api.kate_div(polys, x)
        assert np.array_equal(rem, ev)
        c = int(rng.integers(0, n_cols))
        want_q, want_r = _kate_div_ints(O.fr_to_ints(polys[c]), xi, R)
        assert O.fr_to_ints(q[c]) == want_q and O.fr_to_ints(rem[c].reshape(1, 4))[0] == want_r, (case, n)
        v = O.random_fr(rng, 1)[0]
        acc = np.zeros((n, 4), dtype=np.uint64)
        for j in range(n_cols):
            acc = O.fr_add(O.fr_mul(acc, np.tile(v, (n, 1))), polys[j])
        assert np.array_equal(api.poly_lincomb(polys, v), acc), (case, n, n_cols)
        den = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
        assert np.array_equal(api.grand_product(polys, den), O.grand_product(polys, den)), (case, n, n_cols)
