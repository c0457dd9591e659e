"""ORACLE — TEST INFRASTRUCTURE ONLY.  A second prover: `create_proof` composed on the CPU from the oracle's bricks.

What the reference reaches through `gen_snark_shplonk` (/root/reference/src/scaffold/mod.rs:296) and `keygen_pk` (:273) —
halo2's keygen and prover, un-vendored third-party code [UPSTREAM-RECALL throughout: parity unpinned, SURVEY 8c] — restated
from the protocol, not from halo2_vectordb_amd/rounds.py's code: textbook where the GPU path is clever.

    GPU path (rounds.py)                                   here
    -----------------------------------------------------  ---------------------------------------------------------------
    quotient on 3 (2) cosets of the 2^k-th roots, the       numerator on ALL 4 n points of the extended domain, one division
    gates on 2, groups folded separately, Vandermonde       by X^n - 1 there, one inverse transform: h's 4 n coefficients,
    recombination of the residues                           cut into pieces (the top piece must come out zero)
    cosets of beta sigma, packed mapping, block streaming    every polynomial whole, every coset whole
    permutation cycles by pointer jumping + radix sort       a stable argsort of the same records
    running products per block, chained on the device        one column of n ratios per set, chained on the host
    Poseidon sponge in C (mulx / AVX-512 IFMA builds)        Python integers around the oracle's width-5 permutation
    SHPLONK with device polynomials                          Python integers, synthetic division

Both provers take the blinding scalars as INPUT (halo2 draws them from OsRng, so two reference runs differ: SURVEY 8 f2): with the
same scalars they must write the same proof bytes — tests/test_gpu_cpu_prover.py compares them, and the verifying keys'
commitments, for the cosine k-means circuit, the Merkle circuit (no lookups: degree 3) and BASELINE C1.  `seeded_blinds`
restates the seeded test hook of ProverRounds.prove(seed=...) (numpy default_rng([seed, i]) bytes, 64 per scalar, reduced as
halo2curves Fr::from_u512 does).

The bricks are oracle.py's C restatements (msm_batch, lde_batch, extended_to_coeff, grand_product, eval_polys, lookup_permute,
poseidon_permute, fr_* batch arithmetic); nothing of halo2_vectordb_amd is imported here.
"""
import time

import numpy as np

from . import oracle as O

R = O.R_MOD
N_BLIND = 7          # rows at the end of every advice column the prover fills with random scalars: blinding_factors() + 1 (rounds.py)
MINIMUM_ROWS = 9     # /root/reference/src/scaffold/mod.rs:383
FIXED = ("sel", "sigma", "cst", "table")      # order in which the verifying key's digest absorbs the fixed commitments
T_WIDTH, T_RF, T_RP = 5, 8, 60                # the transcript's sponge: Poseidon x^5, width 5, 8 + 60 rounds (poseidonperm_x5_254_5)


# --------------------------------------------------------------------------------------------------------------- field helpers
def _m(v):
    """canonical int -> one Montgomery element (4,)"""
    return O.fr_from_ints([v % R])[0]


def _full(v, n):
    """canonical int -> (n, 4) array of that element"""
    return np.ascontiguousarray(np.broadcast_to(_m(v), (n, 4)))


def _scale(a, v):
    return O.fr_mul(a, _full(v, a.shape[0]))


def _addc(a, v):
    return O.fr_add(a, _full(v, a.shape[0]))


def _ints(a):
    return O.fr_to_ints(np.ascontiguousarray(a).reshape(-1, 4))


# --------------------------------------------------------------------------------------------------------------- transcript
class Sponge:
    """The Fiat–Shamir transcript as the library documents it (include/vdb.h vdb_transcript_*; restated in tests/test_transcript_cpu.py
    `_sponge`): absorbed values are buffered; a squeeze absorbs the buffer RATE = 4 values per permutation, adds one to the word after
    a short chunk, runs an extra permutation when the buffer's length is a multiple of RATE, returns word 1; the state persists.
    A point enters as its affine x and y reduced mod r (the identity as 0, 0) and is written to the proof compressed: x in 254
    bits little-endian, bit 254 = y odd.  A scalar is written as 32 little-endian bytes."""

    def __init__(self):
        self.state = [1 << 64] + [0] * (T_WIDTH - 1)
        self.buf = []
        self.proof = bytearray()

    def _permute(self):
        st = O.fr_from_ints(self.state)
        self.state = O.fr_to_ints(O.poseidon_permute(st, optimized=False, t=T_WIDTH, r_f=T_RF, r_p=T_RP))

    def common_scalar(self, v):
        self.buf.append(int(v) % R)

    def common_point(self, pt):
        x, y = O.fq_to_ints(np.asarray(pt, dtype=np.uint64).reshape(2, 4))
        self.buf += [x % R, y % R]
        return x, y

    def write_scalar(self, v):
        self.common_scalar(v)
        self.proof += (int(v) % R).to_bytes(32, "little")

    def write_point(self, pt):
        x, y = self.common_point(pt)
        self.proof += (x | ((y & 1) << 254 if (x or y) else 0)).to_bytes(32, "little")

    def write_points(self, pts):
        for pt in np.asarray(pts, dtype=np.uint64).reshape(-1, 8):
            self.write_point(pt)

    def squeeze(self):
        rate = T_WIDTH - 1
        msg, self.buf = self.buf, []
        chunks = [msg[i:i + rate] for i in range(0, len(msg), rate)]
        if len(msg) % rate == 0:
            chunks.append([])
        for ch in chunks:
            for i, v in enumerate(ch):
                self.state[1 + i] = (self.state[1 + i] + v) % R
            if len(ch) < rate:
                self.state[1 + len(ch)] = (self.state[1 + len(ch)] + 1) % R
            self._permute()
        return self.state[1]


# --------------------------------------------------------------------------------------------------------------- circuit, keygen
class Circuit:
    """What keygen is given (a description of the circuit, no witness): the flat advice stream's shape as halo2-base lays it out.
    k, lookup_bits; break_points (rows per advice column; the cell that ends column c starts column c + 1 again);
    gate[i]: a vertical gate q (a + b c - d) starts at stream cell i; n_lookup lookup cells in columns of 2^k - MINIMUM_ROWS;
    copy_of[i] (the earlier cell stream cell i copies, or i), const_idx[i] (-1, or the row of the constants' fixed column the
    cell is tied to: `consts`, canonical integers), lookup_src[j] (the advice cell lookup cell j copies), instance_cells (the
    stream cells the closure makes public, in make_public order: src/scaffold/mod.rs:378-400)."""

    def __init__(self, k, lookup_bits, break_points, gate, n_lookup, copy_of, const_idx, consts, lookup_src, instance_cells):
        self.k, self.L, self.rows = int(k), int(lookup_bits), 1 << int(k)
        self.bp = np.asarray(break_points, dtype=np.int64)
        self.gate = np.asarray(gate).astype(bool)
        self.n_cells, self.n_lookup = int(self.gate.size), int(n_lookup)
        self.copy_of = np.asarray(copy_of, dtype=np.int64)
        self.const_idx = np.asarray(const_idx, dtype=np.int64)
        self.consts = [int(v) % R for v in consts]
        self.lookup_src = np.asarray(lookup_src if lookup_src is not None else [], dtype=np.int64)
        self.instance_cells = [int(c) for c in instance_cells]
        self.n_adv = self.bp.size + 1
        self.lookup_rows = self.rows - MINIMUM_ROWS
        self.n_lk = -(-self.n_lookup // self.lookup_rows)
        self.n_cols = self.n_adv + self.n_lk
        self.n_perm = self.n_cols + 2                    # [advice | lookup | constants | instance]
        # halo2 ConstraintSystem::degree(): 4 with a lookup argument, 3 without (gates q (a + b c - d): 3, permutation: 3)
        self.degree = 4 if self.n_lk else 3
        self.chunk_len, self.n_h = self.degree - 2, self.degree - 1
        self.n_sets = -(-self.n_perm // self.chunk_len)
        self.usable = self.rows - N_BLIND
        self.starts = np.concatenate([[0], np.cumsum(self.bp)]).astype(np.int64)
        assert self.copy_of.size == self.n_cells == self.const_idx.size and self.lookup_src.size in (0, self.n_lookup)
        assert len(self.consts) <= self.usable and len(self.instance_cells) <= self.usable


def layout_advice(cs, stream, blinds=None):
    """stream (n_cells, 4) -> (n_adv, rows, 4): column c = stream[starts[c] : starts[c] + bp[c] + 1], rows N_BLIND from the end = blinds[c]"""
    cols = O.layout_columns(stream, cs.bp.astype(np.uint64), cs.k, cs.n_adv)
    assert cols.shape[0] == cs.n_adv
    if blinds is not None:
        cols[:, cs.rows - N_BLIND:] = blinds
    return cols


def layout_lookup(cs, lookup, blinds=None):
    cols = O.layout_lookup(lookup, cs.k, max(cs.n_lk, 1), MINIMUM_ROWS)[: cs.n_lk] if cs.n_lk else np.zeros((0, cs.rows, 4), dtype=np.uint64)
    if blinds is not None and cs.n_lk:
        cols[:, cs.rows - N_BLIND:] = blinds
    return cols


def layout_selectors(cs):
    """q[c][row] = 1 where a gate starts; the cell a column ends on starts the next column again and opens its gate there only"""
    q = np.zeros((cs.n_adv, cs.rows), dtype=bool)
    for c in range(cs.n_adv):
        lo = int(cs.starts[c])
        n = int(cs.bp[c]) if c < cs.bp.size else cs.n_cells - lo
        q[c, :n] = cs.gate[lo: lo + n]
    one = _m(1)
    out = np.zeros((cs.n_adv, cs.rows, 4), dtype=np.uint64)
    out[q] = one
    return out


def permutation_mapping(cs):
    """The permutation's cycles over the grid [advice | lookup | constants | instance] x rows as words col << 32 | row (identity
    where nothing is tied).  One record per grid position of a copy class, in this order: the stream cells, the overlap cells
    (stream cell starts[c + 1] once more, at row bp[c] of column c), the lookup cells, the used rows of the constants' column, the
    used rows of the instance column; a record's class is the root of its cell under copy_of (a cell tied to constant r has the
    fixed cell r as its root); inside a class every record maps to the next one in record order, the last to the first
    (include/vdb.h vdb_permutation_mapping_dev states the same rule; halo2's own cycle order is [UPSTREAM-RECALL], unpinned)."""
    n_cells, rows = cs.n_cells, cs.rows
    root = np.concatenate([cs.copy_of, n_cells + np.arange(len(cs.consts), dtype=np.int64)])
    tied = np.flatnonzero(cs.const_idx >= 0)
    if (cs.copy_of[tied] != tied).any():
        raise ValueError("a cell tied to a constant must be the root of its copies")
    root[tied] = n_cells + cs.const_idx[tied]
    while True:
        nxt = root[root]
        if np.array_equal(nxt, root):
            break
        root = nxt
    s = np.arange(n_cells, dtype=np.int64)
    col = np.searchsorted(cs.starts, s, side="right") - 1
    rec_root, rec_col, rec_row = [root[:n_cells]], [col], [s - cs.starts[col]]
    d = np.arange(cs.bp.size, dtype=np.int64)
    rec_root.append(root[cs.starts[1:]]), rec_col.append(d), rec_row.append(cs.bp)
    if cs.lookup_src.size:
        j = np.arange(cs.n_lookup, dtype=np.int64)
        rec_root.append(root[cs.lookup_src]), rec_col.append(cs.n_adv + j // cs.lookup_rows), rec_row.append(j % cs.lookup_rows)
    r = np.arange(len(cs.consts), dtype=np.int64)
    rec_root.append(n_cells + r), rec_col.append(np.full(r.size, cs.n_cols)), rec_row.append(r)
    i = np.arange(len(cs.instance_cells), dtype=np.int64)
    rec_root.append(root[np.asarray(cs.instance_cells, dtype=np.int64)] if i.size else i), rec_col.append(np.full(i.size, cs.n_cols + 1)), rec_row.append(i)
    rec_root, rec_col, rec_row = (np.concatenate(a).astype(np.int64) for a in (rec_root, rec_col, rec_row))
    order = np.argsort(rec_root, kind="stable")
    kr, kc, kw = rec_root[order], rec_col[order], rec_row[order]
    first = np.concatenate([[True], kr[1:] != kr[:-1]])
    start = np.maximum.accumulate(np.where(first, np.arange(kr.size), 0))
    last = np.concatenate([kr[1:] != kr[:-1], [True]])
    nxt = np.where(last, start, np.arange(kr.size) + 1)
    mapping = (np.arange(cs.n_perm, dtype=np.uint64)[:, None] << np.uint64(32)) | np.arange(rows, dtype=np.uint64)[None, :]
    mapping[kc, kw] = (kc[nxt].astype(np.uint64) << np.uint64(32)) | kw[nxt].astype(np.uint64)
    return mapping


class ProvingKey:
    pass


def keygen(cs, g_monomial, g_lagrange, threads=4):
    """keygen_vk / keygen_pk (reached from src/scaffold/mod.rs:273): the fixed polynomials — gate selectors, sigma columns, the
    constants' column (constant r at row r), the range table 0 .. 2^L - 1 — in Lagrange and coefficient form, their commitments
    (Lagrange basis), and the key's one-scalar digest: the squeeze of a sponge of its own over the fixed commitments in FIXED order."""
    pk = ProvingKey()
    pk.cs, pk.g, pk.gl, pk.threads = cs, g_monomial, g_lagrange, threads
    n = cs.rows
    lag = {"sel": layout_selectors(cs)}
    pk.mapping = permutation_mapping(cs)
    lag["sigma"] = O.permutation_sigma(pk.mapping, cs.k)
    cst = np.zeros((1, n, 4), dtype=np.uint64)
    if cs.consts:
        cst[0, : len(cs.consts)] = O.fr_from_ints(cs.consts)
    lag["cst"] = cst
    tab = [v if v < (1 << cs.L) else 0 for v in range(n)]
    lag["table"] = O.fr_from_ints(tab).reshape(1, n, 4)
    pk.lag, pk.coeff, pk.commits = lag, {}, {}
    for name in FIXED:
        pk.commits[name] = O.msm_batch(lag[name], g_lagrange, threads=threads)
        pk.coeff[name] = np.stack([O.lagrange_to_coeff(c) for c in lag[name]]) if len(lag[name]) else lag[name]
    tr = Sponge()
    for name in FIXED:
        for pt in pk.commits[name]:
            tr.common_point(pt)
    pk.vk_digest = tr.squeeze()
    # l_0, l_last (the last usable row), l_active (the rows before it): the verifier's own, never committed
    sel = np.zeros((3, n, 4), dtype=np.uint64)
    one = _m(1)
    sel[0, 0], sel[1, cs.usable], sel[2, : cs.usable] = one, one, one
    pk.lagsel_coeff = np.stack([O.lagrange_to_coeff(c) for c in sel])
    return pk


# --------------------------------------------------------------------------------------------------------------- blinding
def wide_scalars(raw, m):
    """m uniform field elements from 64 m bytes: each 512-bit little-endian integer mod r (halo2curves Fr::from_u512 / Fr::random)"""
    assert len(raw) == 64 * m
    return O.fr_from_ints([int.from_bytes(raw[64 * i: 64 * i + 64], "little") % R for i in range(m)])


def seeded_blinds(cs, seed):
    """The blinding scalars ProverRounds.prove(seed=S) uses (its reproducible test hook; a real proof draws them from the OS):
    stream i = numpy default_rng([S, i]).bytes(64 m) through wide_scalars — i = 0: N_BLIND per advice / lookup column; 1, 2: N_BLIND
    per permuted input / table column; 3: N_BLIND - 1 per permutation product; 4: N_BLIND - 1 per lookup product; 5: the n
    coefficients of the vanishing argument's random polynomial."""
    def draw(i, m):
        return wide_scalars(np.random.default_rng([int(seed), i]).bytes(64 * m), m)
    nb = N_BLIND
    return {"adv": draw(0, cs.n_cols * nb).reshape(cs.n_cols, nb, 4), "pa": draw(1, cs.n_lk * nb).reshape(cs.n_lk, nb, 4),
            "ps": draw(2, cs.n_lk * nb).reshape(cs.n_lk, nb, 4), "zp": draw(3, cs.n_sets * (nb - 1)).reshape(cs.n_sets, nb - 1, 4),
            "zl": draw(4, cs.n_lk * (nb - 1)).reshape(cs.n_lk, nb - 1, 4), "rand": draw(5, cs.rows)}


# --------------------------------------------------------------------------------------------------------------- the prover
def _coeffs(cols):
    return np.stack([O.lagrange_to_coeff(c) for c in cols]) if len(cols) else np.zeros((0,) + cols.shape[1:], dtype=np.uint64)


class _Ext:
    """evaluations on the extended domain: 4 n points zeta w_4n^j, j natural order; a rotation by r rows is a shift by 4 r points"""

    def __init__(self, k):
        self.k, self.n, self.ne = k, 1 << k, 4 << k

    def of(self, coeff):
        return O.coeff_to_extended(coeff, ext=2)

    def rot(self, e, r):
        return np.ascontiguousarray(np.roll(e, -4 * r, axis=0))


def prove(pk, stream, lookup, blinds, instances=None, timings=None):
    """create_proof for one circuit instance.  stream (n_cells, 4), lookup (n_lookup, 4): the witness as halo2-base's Context holds
    it; blinds: seeded_blinds' dictionary; instances: canonical integers, None = the public cells' own values.
    Returns dict(proof bytes, instances, challenges, commitments)."""
    cs, th = pk.cs, pk.threads
    n, k, usable, ne = cs.rows, cs.k, cs.usable, 4 * cs.rows
    T = {} if timings is None else timings
    t_last = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        T[name] = T.get(name, 0.0) + now - t_last[0]
        t_last[0] = now

    X = _Ext(k)
    w = O.fr_to_ints(O.root_of_unity(k).reshape(1, 4))[0]
    delta = O.DELTA_INT
    tr = Sponge()
    tr.common_scalar(pk.vk_digest)
    # ---- the statement and the advice round
    stream = np.ascontiguousarray(stream, dtype=np.uint64)
    if instances is None:
        instances = _ints(stream[cs.instance_cells]) if cs.instance_cells else []
    instances = [int(v) % R for v in instances]
    assert len(instances) == len(cs.instance_cells)
    for v in instances:
        tr.common_scalar(v)
    adv_lag = layout_advice(cs, stream, blinds["adv"][: cs.n_adv])
    lk_lag = layout_lookup(cs, lookup, blinds["adv"][cs.n_adv:])
    cols_lag = np.concatenate([adv_lag, lk_lag])                       # [advice | lookup]: the committed witness columns
    C = {"adv": O.msm_batch(cols_lag, pk.gl, threads=th)}
    tr.write_points(C["adv"])
    tr.squeeze()                                                       # theta: squeezed as halo2 does, unused (single-column lookups)
    lap("advice")
    # ---- the lookup argument's permuted columns (halo2 plonk/lookup/prover.rs permute_expression_pair, over the usable rows)
    table_int = _ints(pk.lag["table"][0])
    pa = np.zeros((cs.n_lk, n, 4), dtype=np.uint64)
    ps = np.zeros((cs.n_lk, n, 4), dtype=np.uint64)
    for c in range(cs.n_lk):
        a, s = O.lookup_permute(_ints(lk_lag[c, :usable]), table_int[:usable])
        pa[c, :usable], ps[c, :usable] = O.fr_from_ints(a), O.fr_from_ints(s)
        pa[c, usable:], ps[c, usable:] = blinds["pa"][c], blinds["ps"][c]
    C["pa"], C["ps"] = O.msm_batch(pa, pk.gl, threads=th), O.msm_batch(ps, pk.gl, threads=th)
    for c in range(cs.n_lk):
        tr.write_point(C["pa"][c])
        tr.write_point(C["ps"][c])
    beta, gamma = tr.squeeze(), tr.squeeze()
    lap("lookup_permute")
    # ---- running products.  Permutation: per set of chunk_len columns z[0] = the previous set's z[usable] (1 for the first),
    # z[i+1] = z[i] prod_c (v_c[i] + beta delta^c w^i + gamma) / (v_c[i] + beta sigma_c[i] + gamma) for i < usable
    inst_lag = np.zeros((1, n, 4), dtype=np.uint64)
    if instances:
        inst_lag[0, : len(instances)] = O.fr_from_ints(instances)
    perm_lag = np.concatenate([cols_lag, pk.lag["cst"], inst_lag])     # the permutation's columns
    wpow = [1] * n
    for i in range(1, n):
        wpow[i] = wpow[i - 1] * w % R
    wp = O.fr_from_ints(wpow)
    one_col = _full(1, n)
    zp = np.zeros((cs.n_sets, n, 4), dtype=np.uint64)
    carry = 1
    for s_i in range(cs.n_sets):
        num, den = one_col.copy(), one_col.copy()
        for c in range(s_i * cs.chunk_len, min((s_i + 1) * cs.chunk_len, cs.n_perm)):
            v = perm_lag[c]
            num = O.fr_mul(num, _addc(O.fr_add(v, _scale(wp, beta * pow(delta, c, R) % R)), gamma))
            den = O.fr_mul(den, _addc(O.fr_add(v, _scale(pk.lag["sigma"][c], beta)), gamma))
        num[usable:], den[usable:] = one_col[usable:], one_col[usable:]
        z = _scale(O.grand_product(num[None], den[None])[0], carry)
        carry = _ints(z[usable])[0]
        z[usable + 1:] = blinds["zp"][s_i]
        zp[s_i] = z
    # Lookup: z[i+1] = z[i] (A + beta)(S + gamma) / ((A' + beta)(S' + gamma))
    zl = np.zeros((cs.n_lk, n, 4), dtype=np.uint64)
    for c in range(cs.n_lk):
        num = O.fr_mul(_addc(lk_lag[c], beta), _addc(pk.lag["table"][0], gamma))
        den = O.fr_mul(_addc(pa[c], beta), _addc(ps[c], gamma))
        num[usable:], den[usable:] = one_col[usable:], one_col[usable:]
        z = O.grand_product(num[None], den[None])[0]
        z[usable + 1:] = blinds["zl"][c]
        zl[c] = z
    C["zp"], C["zl"] = O.msm_batch(zp, pk.gl, threads=th), O.msm_batch(zl, pk.gl, threads=th)
    tr.write_points(C["zp"])
    tr.write_points(C["zl"])
    # the vanishing argument's random polynomial (halo2 plonk/vanishing/prover.rs): n uniform coefficients, committed before y
    rand_c = np.ascontiguousarray(blinds["rand"]).reshape(1, n, 4)
    C["rand"] = O.msm_batch(rand_c, pk.g, threads=1)
    tr.write_points(C["rand"])
    y = tr.squeeze()
    lap("products")
    # ---- the quotient: numerator = sum_i term_i y^(N-1-i) on the 4 n points, terms in this order: one gate per advice column;
    # l_0 (1 - z_0); l_last (z_last^2 - z_last); l_0 (z_i - z_{i-1}(w^-N_BLIND X)) for i >= 1; per set l_active (z_i(wX) prod (v + beta
    # sigma + gamma) - z_i(X) prod (v + beta delta^c X + gamma)); per lookup column l_0 (1 - Z), l_last (Z^2 - Z),
    # l_active (Z(wX)(A' + beta)(S' + gamma) - Z (A + beta)(S + gamma)), l_0 (A' - S'), l_active (A' - S')(A' - A'(w^-1 X))
    co = {"adv": _coeffs(cols_lag), "pa": _coeffs(pa), "ps": _coeffs(ps), "zp": _coeffs(zp), "zl": _coeffs(zl), "rand": rand_c}
    inst_coeff = _coeffs(inst_lag)
    perm_coeff = np.concatenate([co["adv"], pk.coeff["cst"], inst_coeff])
    ext_perm = [X.of(c) for c in perm_coeff]
    l0, ll, la = (X.of(c) for c in pk.lagsel_coeff)
    zeta = O.fr_to_ints(O.zeta().reshape(1, 4))[0]
    w4n = O.fr_to_ints(O.root_of_unity(k + 2).reshape(1, 4))[0]
    xs = [zeta] * ne
    for j in range(1, ne):
        xs[j] = xs[j - 1] * w4n % R
    x_ext = O.fr_from_ints(xs)                                         # the extended domain's points themselves
    acc = np.zeros((ne, 4), dtype=np.uint64)

    def fold(term):
        nonlocal acc
        acc = O.fr_add(_scale(acc, y), term)

    for c in range(cs.n_adv):
        a = ext_perm[c]
        fold(O.fr_mul(X.of(pk.coeff["sel"][c]), O.fr_sub(O.fr_add(a, O.fr_mul(X.rot(a, 1), X.rot(a, 2))), X.rot(a, 3))))
    ez = [X.of(c) for c in co["zp"]]
    one_e = _full(1, ne)
    fold(O.fr_mul(l0, O.fr_sub(one_e, ez[0])))
    fold(O.fr_mul(ll, O.fr_sub(O.fr_mul(ez[-1], ez[-1]), ez[-1])))
    for s_i in range(1, cs.n_sets):
        fold(O.fr_mul(l0, O.fr_sub(ez[s_i], X.rot(ez[s_i - 1], -N_BLIND))))
    for s_i in range(cs.n_sets):
        left, right = X.rot(ez[s_i], 1), ez[s_i]
        for c in range(s_i * cs.chunk_len, min((s_i + 1) * cs.chunk_len, cs.n_perm)):
            left = O.fr_mul(left, _addc(O.fr_add(ext_perm[c], _scale(X.of(pk.coeff["sigma"][c]), beta)), gamma))
            right = O.fr_mul(right, _addc(O.fr_add(ext_perm[c], _scale(x_ext, beta * pow(delta, c, R) % R)), gamma))
        fold(O.fr_mul(la, O.fr_sub(left, right)))
    e_tab = X.of(pk.coeff["table"][0])
    for c in range(cs.n_lk):
        A, PA, PS, Z = ext_perm[cs.n_adv + c], X.of(co["pa"][c]), X.of(co["ps"][c]), X.of(co["zl"][c])
        fold(O.fr_mul(l0, O.fr_sub(one_e, Z)))
        fold(O.fr_mul(ll, O.fr_sub(O.fr_mul(Z, Z), Z)))
        fold(O.fr_mul(la, O.fr_sub(O.fr_mul(X.rot(Z, 1), O.fr_mul(_addc(PA, beta), _addc(PS, gamma))),
                                   O.fr_mul(Z, O.fr_mul(_addc(A, beta), _addc(e_tab, gamma))))))
        d = O.fr_sub(PA, PS)
        fold(O.fr_mul(l0, d))
        fold(O.fr_mul(la, O.fr_mul(d, O.fr_sub(PA, X.rot(PA, -1)))))
    # h = numerator / (X^n - 1): on the coset X^n - 1 takes four values, zeta^n i^j - 1
    zn, w4 = pow(zeta, n, R), pow(w4n, n, R)
    inv4 = [pow((zn * pow(w4, j, R) - 1) % R, -1, R) for j in range(4)]
    acc = O.fr_mul(acc, np.ascontiguousarray(np.tile(O.fr_from_ints(inv4), (ne // 4, 1))))
    h_all = O.extended_to_coeff(acc[None], k, ext=2)[0]
    if h_all[cs.n_h * n:].any():
        raise ValueError("the quotient has degree >= (degree - 1) n: the witness does not satisfy the circuit")
    h = h_all[: cs.n_h * n].reshape(cs.n_h, n, 4)
    C["h"] = O.msm_batch(h, pk.g, threads=th)
    tr.write_points(C["h"])
    x = tr.squeeze()
    lap("quotient")
    # h folded at x: hf = sum_i x^(n i) h_i, opened beside the random polynomial; neither its commitment nor its value is sent
    xn = pow(x, n, R)
    hf = h[0].copy()
    for i in range(1, cs.n_h):
        hf = O.fr_add(hf, _scale(h[i], pow(xn, i, R)))
    co["hf"] = hf.reshape(1, n, 4)
    C["hf"] = O.msm_batch(co["hf"], pk.g, threads=1)
    # ---- evaluations: which polynomial is read at which rotation (gate columns at rows 0..3, products one row ahead, the permuted
    # input one row back, the chained product N_BLIND rows back); halo2 opens a column at the rotations its queries name
    polys = dict(co)
    polys["advg"] = co["adv"][: cs.n_adv]
    for name in FIXED:
        polys[name] = pk.coeff[name]
    opened = {0: ["adv", "sel", "sigma", "cst", "table", "pa", "ps", "zp", "zl", "hf", "rand"], 1: ["advg", "zp", "zl"], 2: ["advg"], 3: ["advg"], -1: ["pa"],
              -N_BLIND: ["zp"]}
    opened = {rot: [nm for nm in names if len(polys[nm])] for rot, names in opened.items()}
    opened = {rot: names for rot, names in opened.items() if names}
    points = {rot: x * pow(w, rot % n, R) % R for rot in opened}
    evals = {}
    for rot, names in opened.items():
        for nm in names:
            evals[(nm, rot)] = _ints(O.eval_polys(polys[nm], _m(points[rot])))
            if nm != "hf":
                for e in evals[(nm, rot)]:
                    tr.write_scalar(e)
    lap("evaluations")
    # ---- SHPLONK (halo2 poly/kzg/multiopen/shplonk): polynomials opened at the same set of points form a rotation set
    by_poly = {}
    for rot, names in opened.items():
        for nm in names:
            by_poly.setdefault(nm, []).append(rot)
    sets = []
    for nm, rots in by_poly.items():
        key = tuple(sorted(rots))
        for sset in sets:
            if sset[0] == key:
                sset[1].append(nm)
                break
        else:
            sets.append((key, [nm]))
    yo, v = tr.squeeze(), tr.squeeze()
    m = len(sets)

    def horner_polys(names, ch):
        accp = np.zeros((n, 4), dtype=np.uint64)
        for nm in names:
            for p in polys[nm]:
                accp = O.fr_add(_scale(accp, ch), p)
        return _ints(accp)

    def interpolate(pts, vals):
        coeffs = [0] * len(pts)
        for i, (xi, yi) in enumerate(zip(pts, vals)):
            basis, denom = [1], 1
            for j, xj in enumerate(pts):
                if j != i:
                    basis = [(a - xj * b) % R for a, b in zip([0] + basis, basis + [0])]
                    denom = denom * (xi - xj) % R
            sc = yi * pow(denom, -1, R) % R
            coeffs = [(c + sc * b) % R for c, b in zip(coeffs, basis)]
        return coeffs

    def divide(p, a):
        """(quotient, remainder) of p(X) / (X - a), p as a list of n integers (low first)"""
        q = [0] * len(p)
        carry_ = 0
        for i in range(len(p) - 1, -1, -1):
            q[i] = carry_
            carry_ = (p[i] + a * carry_) % R
        return q, carry_

    def at(p, a):
        r_ = 0
        for c in reversed(p):
            r_ = (r_ * a + c) % R
        return r_

    q_polys, r_polys = [], []
    f = [0] * n
    for rots, names in sets:
        q = horner_polys(names, yo)
        vals = []
        for rot in rots:
            e = 0
            for nm in names:
                for ev in evals[(nm, rot)]:
                    e = (e * yo + ev) % R
            vals.append(e)
        r = interpolate([points[rot] for rot in rots], vals)
        q_polys.append(q)
        r_polys.append(r)
        num = list(q)
        for i, c in enumerate(r):
            num[i] = (num[i] - c) % R
        for rot in rots:
            num, rem = divide(num, points[rot])
            if rem:
                raise AssertionError("an evaluation is not the polynomial's value")
        f = [(a * v + b) % R for a, b in zip(f, num)]                 # f = f v + (q_S - r_S) / Z_S
    W1 = O.msm_batch(O.fr_from_ints(f).reshape(1, n, 4), pk.g, threads=1)[0]
    tr.write_point(W1)
    u = tr.squeeze()
    all_rots = sorted({rot for rots, _ in sets for rot in rots})

    def vanish(rots, a):
        r_ = 1
        for rot in rots:
            r_ = r_ * (a - points[rot]) % R
        return r_

    lin = [0] * n
    for s_i, (rots, names) in enumerate(sets):
        coef = pow(v, m - 1 - s_i, R) * vanish([rot for rot in all_rots if rot not in rots], u) % R
        r_u = at(r_polys[s_i], u)
        lin = [(a + coef * b) % R for a, b in zip(lin, q_polys[s_i])]
        lin[0] = (lin[0] - coef * r_u) % R
    zt = vanish(all_rots, u)
    lin = [(a - zt * b) % R for a, b in zip(lin, f)]
    lq, rem = divide(lin, u)
    if rem:
        raise AssertionError("the linearisation polynomial does not vanish at u")
    W2 = O.msm_batch(O.fr_from_ints(lq).reshape(1, n, 4), pk.g, threads=1)[0]
    tr.write_point(W2)
    lap("openings")
    return dict(proof=bytes(tr.proof), instances=instances, challenges=dict(beta=beta, gamma=gamma, y=y, x=x, yo=yo, v=v, u=u), commitments=C,
                opened=opened, evals=evals, W1=W1, W2=W2)
