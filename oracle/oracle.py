"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (the C restatement of the reference's hot path).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (halo2_vectordb_amd) never does.

Field elements travel as numpy uint64 arrays of shape (..., 4): little-endian limbs in Montgomery
form (R = 2^256), the in-memory layout of halo2curves `Fr` that the reference's FFI would hand over.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
MONT_R = 1 << 256


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_init()
        _lib.orc_ctx_new.restype = ctypes.c_void_p
        _lib.orc_ctx_len.restype = ctypes.c_size_t
        _lib.orc_ctx_lookup_len.restype = ctypes.c_size_t
        _lib.orc_ctx_advice.restype = ctypes.c_void_p
        _lib.orc_ctx_lookup.restype = ctypes.c_void_p
        _lib.orc_ctx_selectors.restype = ctypes.c_void_p
        _lib.orc_ctx_break_points.restype = ctypes.c_size_t
        _lib.orc_check_gates.restype = ctypes.c_size_t
        _lib.orc_layout_columns.restype = ctypes.c_size_t
        _lib.orc_layout_lookup.restype = ctypes.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _sz(n):
    return ctypes.c_size_t(int(n))


# ---------------------------------------------------------------- int <-> limb helpers
def ints_to_limbs(vals):
    """list of python ints (< 2^256) -> uint64 array (n, 4), no Montgomery conversion."""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(a[i, k]) << (64 * k) for k in range(4)) for i in range(a.shape[0])]


def fr_from_ints(vals):
    """canonical ints -> Montgomery Fr array (n,4)."""
    return ints_to_limbs([(v % R_MOD) * MONT_R % R_MOD for v in vals])


def fr_to_ints(a):
    rinv = pow(MONT_R, -1, R_MOD)
    return [v * rinv % R_MOD for v in limbs_to_ints(a)]


def fq_from_ints(vals):
    return ints_to_limbs([(v % Q_MOD) * MONT_R % Q_MOD for v in vals])


def fq_to_ints(a):
    rinv = pow(MONT_R, -1, Q_MOD)
    return [v * rinv % Q_MOD for v in limbs_to_ints(a)]


def random_fr(rng, n):
    """uniform-ish Fr elements, Montgomery form; rng = numpy Generator."""
    raw = rng.integers(0, 1 << 63, size=(n, 5), dtype=np.int64).astype(object)
    vals = [(int(r[0]) | (int(r[1]) << 63) | (int(r[2]) << 126) | (int(r[3]) << 189) | (int(r[4]) << 252)) % R_MOD for r in raw]
    return fr_from_ints(vals)


# ---------------------------------------------------------------- field batch ops
def _binop(name, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    o = np.empty_like(a)
    getattr(lib(), name)(_p(o), _p(a), _p(b), _sz(a.size // 4))
    return o


def fr_mul(a, b):
    return _binop("orc_fr_mul_batch", a, b)


def fr_add(a, b):
    return _binop("orc_fr_add_batch", a, b)


def fr_sub(a, b):
    return _binop("orc_fr_sub_batch", a, b)


def fr_inv(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    o = np.empty_like(a)
    lib().orc_fr_inv_batch(_p(o), _p(a), _sz(a.size // 4))
    return o


def eval_polys(coeffs, x):
    """out[c] = sum_i coeffs[c][i] * x^i (Horner); coeffs: (n_cols, n, 4), x: (4,)"""
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.zeros((coeffs.shape[0], 4), dtype=np.uint64)
    for c in range(coeffs.shape[0]):
        lib().orc_eval_poly(_p(out[c]), _p(coeffs[c]), _sz(coeffs.shape[1]), _p(x))
    return out


def lookup_permute(input_vals, table_vals):
    """halo2 plonk/lookup/prover.rs `permute_expression_pair` over the usable rows, restated literally on canonical
    integers ([UPSTREAM-RECALL]): sorted input; the table value at the first row of each run, the left-over table values
    (ascending) popped onto the repeated rows from the last one backwards.  Returns (permuted_input, permuted_table) as
    lists of ints; raises ValueError where upstream panics (an input value that the table does not hold)."""
    a = sorted(int(v) for v in input_vals)
    left = {}
    for v in table_vals:
        left[int(v)] = left.get(int(v), 0) + 1
    s = [0] * len(a)
    repeated = []
    for row, v in enumerate(a):
        if row == 0 or v != a[row - 1]:
            s[row] = v
            if left.get(v, 0) == 0:
                raise ValueError("input value not in table")
            left[v] -= 1
        else:
            repeated.append(row)
    for v in sorted(left):
        for _ in range(left[v]):
            s[repeated.pop()] = v
    if repeated:
        raise ValueError("table smaller than the input")
    return a, s


def grand_product(num, den):
    """z[c][0] = 1, z[c][i+1] = z[c][i] * num[c][i] / den[c][i] per column (zero denominators invert to zero)."""
    num = np.ascontiguousarray(num, dtype=np.uint64)
    den = np.ascontiguousarray(den, dtype=np.uint64)
    z = np.empty_like(num)
    for c in range(num.shape[0]):
        lib().orc_grand_product(_p(z[c]), _p(num[c]), _p(den[c]), _sz(num.shape[1]))
    return z


DELTA_INT = pow(7, 1 << 28, R_MOD)      # halo2curves bn256 Fr::DELTA = GENERATOR^(2^S)  [UPSTREAM-RECALL]


def permutation_sigma(mapping, k, delta=DELTA_INT):
    """halo2 plonk/permutation/keygen.rs: sigma[c][row] = delta^c' omega^row' where mapping[c][row] = c' << 32 | row'."""
    w = fr_to_ints(root_of_unity(k).reshape(1, 4))[0]
    n_cols, n = mapping.shape
    wp = [1] * n
    for i in range(1, n):
        wp[i] = wp[i - 1] * w % R_MOD
    dp = [pow(delta, c, R_MOD) for c in range(n_cols)]
    vals = [dp[int(m) >> 32] * wp[int(m) & 0xFFFFFFFF] % R_MOD for m in mapping.reshape(-1)]
    return fr_from_ints(vals).reshape(n_cols, n, 4)


def permutation_product(cols, sigma, usable_rows, chunk_len, beta, gamma, delta=DELTA_INT):
    """halo2 plonk/permutation/prover.rs commit ([UPSTREAM-RECALL]), on canonical integers, one inversion per row: per
    chunk of chunk_len columns z[i+1] = z[i] prod_c (v + delta^c beta omega^i + gamma) / (v + beta sigma + gamma); each
    chunk starts from the previous chunk's z[usable_rows]; rows above usable_rows are zero."""
    n_cols, n = cols.shape[0], cols.shape[1]
    k = n.bit_length() - 1
    w = fr_to_ints(root_of_unity(k).reshape(1, 4))[0]
    v = np.array(fr_to_ints(cols.reshape(-1, 4)), dtype=object).reshape(n_cols, n)
    sg = np.array(fr_to_ints(sigma.reshape(-1, 4)), dtype=object).reshape(n_cols, n)
    b, g = fr_to_ints(beta.reshape(1, 4))[0], fr_to_ints(gamma.reshape(1, 4))[0]
    out = []
    last = 1
    for c0 in range(0, n_cols, chunk_len):
        z = [0] * n
        z[0] = last
        wi = 1
        for i in range(usable_rows):
            num = den = 1
            for c in range(c0, min(c0 + chunk_len, n_cols)):
                num = num * (v[c][i] + pow(delta, c, R_MOD) * b % R_MOD * wi + g) % R_MOD
                den = den * (v[c][i] + b * sg[c][i] + g) % R_MOD
            z[i + 1] = z[i] * num % R_MOD * pow(den, -1, R_MOD) % R_MOD
            wi = wi * w % R_MOD
        last = z[usable_rows]
        out.append(fr_from_ints(z))
    return np.stack(out)


def lookup_product(inputs, table, perm_inputs, perm_table, usable_rows, beta, gamma):
    """halo2 plonk/lookup/prover.rs commit_product ([UPSTREAM-RECALL]): z[i+1] = z[i] (A + beta)(S + gamma) /
    ((A' + beta)(S' + gamma)) for i < usable_rows, rows above zero."""
    n_cols, n = inputs.shape[0], inputs.shape[1]
    b, g = fr_to_ints(beta.reshape(1, 4))[0], fr_to_ints(gamma.reshape(1, 4))[0]
    S = fr_to_ints(table.reshape(-1, 4))
    out = []
    for c in range(n_cols):
        A, PA, PS = fr_to_ints(inputs[c]), fr_to_ints(perm_inputs[c]), fr_to_ints(perm_table[c])
        z = [0] * n
        z[0] = 1
        for i in range(usable_rows):
            den = (PA[i] + b) * (PS[i] + g) % R_MOD
            z[i + 1] = z[i] * ((A[i] + b) * (S[i] + g) % R_MOD) % R_MOD * pow(den, -1, R_MOD) % R_MOD
        out.append(fr_from_ints(z))
    return np.stack(out)


def fr_to_canonical(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    o = np.empty_like(a)
    lib().orc_fr_to_canonical_batch(_p(o), _p(a), _sz(a.size // 4))
    return o


def fr_from_canonical(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    o = np.empty_like(a)
    lib().orc_fr_from_canonical_batch(_p(o), _p(a), _sz(a.size // 4))
    return o


def root_of_unity(k):
    o = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_root_of_unity(_p(o), ctypes.c_uint(k))
    return o


def zeta():
    o = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_zeta(_p(o))
    return o


# ---------------------------------------------------------------- G1 / MSM / NTT
def g1_generator():
    o = np.zeros(8, dtype=np.uint64)
    lib().orc_g1_generator(_p(o))
    return o


def g1_mul_generator(scalars_canonical_ints):
    s = ints_to_limbs([v % R_MOD for v in scalars_canonical_ints])
    o = np.zeros((len(scalars_canonical_ints), 8), dtype=np.uint64)
    lib().orc_g1_mul_generator_batch(_p(o), _p(s), _sz(len(scalars_canonical_ints)))
    return o


def srs_from_tau(k, tau):
    n = 1 << k
    g = np.zeros((n, 8), dtype=np.uint64)
    gl = np.zeros((n, 8), dtype=np.uint64)
    t = ints_to_limbs([tau % R_MOD])
    lib().orc_srs_from_tau(_p(g), _p(gl), ctypes.c_uint(k), _p(t))
    return g, gl


def msm_naive(scalars, bases):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    o = np.zeros(8, dtype=np.uint64)
    lib().orc_msm_naive(_p(o), _p(scalars), _p(bases), _sz(scalars.size // 4))
    return o


def msm(scalars, bases, threads=1):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    o = np.zeros(8, dtype=np.uint64)
    lib().orc_msm_pippenger(_p(o), _p(scalars), _p(bases), _sz(scalars.size // 4), ctypes.c_int(threads))
    return o


def msm_batch(cols, bases, threads=1):
    """cols: (n_cols, n, 4) Montgomery scalars; returns (n_cols, 8) affine points."""
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    n_cols, n = cols.shape[0], cols.shape[1]
    o = np.zeros((n_cols, 8), dtype=np.uint64)
    lib().orc_msm_batch(_p(o), _p(cols), _p(bases), _sz(n), _sz(n_cols), ctypes.c_int(threads))
    return o


def ntt(a, omega):
    a = np.array(a, dtype=np.uint64, copy=True)
    n = a.size // 4
    log_n = n.bit_length() - 1
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    lib().orc_ntt(_p(a), ctypes.c_uint(log_n), _p(omega))
    return a


def ntt_naive(a, omega):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    n = a.size // 4
    o = np.empty_like(a)
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    lib().orc_ntt_naive_dft(_p(o), _p(a), ctypes.c_uint(n.bit_length() - 1), _p(omega))
    return o


def lagrange_to_coeff(a):
    a = np.array(a, dtype=np.uint64, copy=True)
    lib().orc_lagrange_to_coeff(_p(a), ctypes.c_uint((a.size // 4).bit_length() - 1))
    return a


def coeff_to_extended(a, ext=2):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    n = a.size // 4
    o = np.zeros((n << ext, 4), dtype=np.uint64)
    lib().orc_coeff_to_extended(_p(o), _p(a), ctypes.c_uint(n.bit_length() - 1), ctypes.c_uint(ext))
    return o


def ntt_batch(cols, omega, threads=1):
    cols = np.array(cols, dtype=np.uint64, copy=True)
    n_cols, n = cols.shape[0], cols.shape[1]
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    lib().orc_ntt_batch(_p(cols), _sz(n_cols), ctypes.c_uint(n.bit_length() - 1), _p(omega), ctypes.c_int(threads))
    return cols


def extended_to_coeff(ext_cols, k, ext=2):
    """inverse of coeff_to_extended per column (all 2^(k+ext) coefficients; the upper ones are zero for a degree < 2^k input)"""
    a = np.array(ext_cols, dtype=np.uint64, copy=True)
    for c in range(a.shape[0]):
        lib().orc_extended_to_coeff(_p(a[c]), ctypes.c_uint(k), ctypes.c_uint(ext))
    return a


def lde_batch(cols, ext=2, threads=1):
    """lagrange_to_coeff then coeff_to_extended for each column; returns (coeffs, extended)."""
    cols = np.array(cols, dtype=np.uint64, copy=True)
    n_cols, n = cols.shape[0], cols.shape[1]
    e = np.zeros((n_cols, n << ext, 4), dtype=np.uint64)
    lib().orc_lde_batch(_p(e), _p(cols), _sz(n_cols), ctypes.c_uint(n.bit_length() - 1), ctypes.c_uint(ext), ctypes.c_int(threads))
    return cols, e


# ---------------------------------------------------------------- Poseidon
def poseidon_permute(state, optimized=False, t=3, r_f=8, r_p=57):
    s = np.array(state, dtype=np.uint64, copy=True)
    fn = lib().orc_poseidon_permute_opt if optimized else lib().orc_poseidon_permute_naive
    fn(t, r_f, r_p, _p(s))
    return s


def poseidon_hash_many(msgs, t=3, r_f=8, r_p=57):
    msgs = np.ascontiguousarray(msgs, dtype=np.uint64)
    n, ln = msgs.shape[0], msgs.shape[1]
    o = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_poseidon_hash_many(t, r_f, r_p, _p(msgs), _sz(n), _sz(ln), _p(o))
    return o


def poseidon_merkle_root(vectors, t=3, r_f=8, r_p=57):
    vectors = np.ascontiguousarray(vectors, dtype=np.uint64)
    n, dim = vectors.shape[0], vectors.shape[1]
    o = np.zeros(4, dtype=np.uint64)
    lib().orc_poseidon_merkle_root(t, r_f, r_p, _p(vectors), _sz(n), _sz(dim), _p(o))
    return o


def poseidon_spec(t=3, r_f=8, r_p=57):
    rc = np.zeros(((r_f + r_p) * t, 4), dtype=np.uint64)
    mds = np.zeros((t * t, 4), dtype=np.uint64)
    lib().orc_poseidon_spec_dump(t, r_f, r_p, _p(rc), _p(mds))
    return rc, mds


# ---------------------------------------------------------------- fixed point / gadgets
def quantize(x, P=48):
    x = np.ascontiguousarray(x, dtype=np.float64)
    o = np.zeros(x.shape + (4,), dtype=np.uint64)
    lib().orc_fp_quantize(ctypes.c_uint(P), _p(x), _p(o), _sz(x.size))
    return o


def dequantize(a, P=48):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    o = np.zeros(a.shape[:-1], dtype=np.float64)
    lib().orc_fp_dequantize(ctypes.c_uint(P), _p(a), _p(o), _sz(a.size // 4))
    return o


OPS = dict(qadd=0, qsub=1, qmul=2, qdiv=3, neg=4, qabs=5, is_neg=6, qmin=7, qsqrt=8, qlog2=9, qexp2=10,
           qlog=11, qexp=12, qpow=13, bit_xor=14, cond_neg=15, signed_div_scale=16, qmax=17, sign=18, clip=19, qmod=20, qsin=21, qcos=22,
           qtan=23, qsinh=24, qcosh=25, qtanh=26)
METRICS = dict(euclidean=0, cosine=1, manhattan=2, hamming=3)


class Ctx:
    """halo2-base Context restated (flat advice stream + cells_to_lookup)."""

    def __init__(self, store=True, keygen=False, plan_k=None, minimum_rows=9):
        self.h = ctypes.c_void_p(lib().orc_ctx_new(int(store), int(keygen)))
        if plan_k is not None:
            lib().orc_ctx_enable_plan(self.h, ctypes.c_uint(plan_k), ctypes.c_uint(minimum_rows))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ctx_free(self.h)
            self.h = None

    def __len__(self):
        return lib().orc_ctx_len(self.h)

    @property
    def n_lookup(self):
        return lib().orc_ctx_lookup_len(self.h)

    @property
    def err(self):
        return lib().orc_ctx_err(self.h)

    def advice(self):
        n = len(self)
        ptr = lib().orc_ctx_advice(self.h)
        if not ptr or n == 0:
            return np.zeros((0, 4), dtype=np.uint64)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(n, 4)).copy()

    def lookup(self):
        n = self.n_lookup
        ptr = lib().orc_ctx_lookup(self.h)
        if not ptr or n == 0:
            return np.zeros((0, 4), dtype=np.uint64)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(n, 4)).copy()

    def selectors(self):
        n = len(self)
        ptr = lib().orc_ctx_selectors(self.h)
        if not ptr or n == 0:
            return np.zeros((0,), dtype=np.uint8)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()

    def break_points(self):
        n = lib().orc_ctx_break_points(self.h, None, _sz(0))
        out = np.zeros(max(n, 1), dtype=np.uint64)
        lib().orc_ctx_break_points(self.h, _p(out), _sz(n))
        return out[:n]

    def check_gates(self, L):
        return lib().orc_check_gates(self.h, ctypes.c_uint(L))

    # ---- emitters
    def assign_witnesses(self, v):
        v = np.ascontiguousarray(v, dtype=np.uint64)
        lib().orc_assign_witnesses(self.h, _p(v), _sz(v.size // 4))

    def op(self, name, a, b=None, P=48, L=13):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b if b is not None else np.zeros(4, dtype=np.uint64), dtype=np.uint64)
        o = np.zeros(4, dtype=np.uint64)
        lib().orc_fp_op(self.h, ctypes.c_uint(P), ctypes.c_uint(L), OPS[name], _p(a), _p(b), _p(o))
        return o

    def distance(self, metric, a, b, P=48, L=13):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        o = np.zeros(4, dtype=np.uint64)
        lib().orc_distance(self.h, ctypes.c_uint(P), ctypes.c_uint(L), METRICS[metric], _p(a), _p(b), _sz(a.size // 4), _p(o))
        return o

    def inner_product(self, a, b, P=48, L=13):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        o = np.zeros(4, dtype=np.uint64)
        lib().orc_inner_product(self.h, ctypes.c_uint(P), ctypes.c_uint(L), _p(a), _p(b), _sz(a.size // 4), _p(o))
        return o

    def nearest_vector(self, metric, query, vectors, P=48, L=13):
        query = np.ascontiguousarray(query, dtype=np.uint64)
        vectors = np.ascontiguousarray(vectors, dtype=np.uint64)
        n, dim = vectors.shape[0], vectors.shape[1]
        ind = np.zeros((n, 4), dtype=np.uint64)
        res = np.zeros((dim, 4), dtype=np.uint64)
        lib().orc_nearest_vector(self.h, ctypes.c_uint(P), ctypes.c_uint(L), METRICS[metric], _p(query), _p(vectors),
                                 _sz(n), _sz(dim), _p(ind), _p(res))
        return ind, res

    def kmeans(self, metric, vectors, K, I, P=48, L=13):
        vectors = np.ascontiguousarray(vectors, dtype=np.uint64)
        n, dim = vectors.shape[0], vectors.shape[1]
        cent = np.zeros((K, dim, 4), dtype=np.uint64)
        ind = np.zeros((n, K, 4), dtype=np.uint64)
        lib().orc_kmeans(self.h, ctypes.c_uint(P), ctypes.c_uint(L), METRICS[metric], _p(vectors), _sz(n), _sz(dim),
                         _sz(K), _sz(I), _p(cent), _p(ind))
        return cent, ind

    def poseidon_chip_new(self, t=3):
        lib().orc_poseidon_chip_new(self.h, t)

    def merkle_commitment(self, vectors, t=3, r_f=8, r_p=57):
        vectors = np.ascontiguousarray(vectors, dtype=np.uint64)
        n, dim = vectors.shape[0], vectors.shape[1]
        o = np.zeros(4, dtype=np.uint64)
        lib().orc_merkle_commitment(self.h, t, r_f, r_p, _p(vectors), _sz(n), _sz(dim), _p(o))
        return o


def layout_columns(stream, break_points, k, n_cols_cap):
    stream = np.ascontiguousarray(stream, dtype=np.uint64)
    bp = np.ascontiguousarray(break_points, dtype=np.uint64)
    cols = np.zeros((n_cols_cap, 1 << k, 4), dtype=np.uint64)
    used = lib().orc_layout_columns(_p(stream), _sz(stream.size // 4), _p(bp), _sz(bp.size), ctypes.c_uint(k), _p(cols), _sz(n_cols_cap))
    return cols[:used]


def layout_lookup(lookup, k, n_cols_cap, minimum_rows=9):
    lookup = np.ascontiguousarray(lookup, dtype=np.uint64)
    cols = np.zeros((n_cols_cap, 1 << k, 4), dtype=np.uint64)
    used = lib().orc_layout_lookup(_p(lookup), _sz(lookup.size // 4), ctypes.c_uint(k), ctypes.c_uint(minimum_rows), _p(cols), _sz(n_cols_cap))
    return cols[:used]
