"""Copy constraints of the fixed-point circuits: for every advice cell that the distance / nearest_vector /
k-means gadgets emit — /root/reference/src/gadget/{fixed_point,distance,vectordb}.rs on the halo2-base GateChip / RangeChip
cell templates — what kind of cell it is and, for copies, which earlier cell it copies.  This is what halo2-base records
while the reference's closure runs (`Existing(cell)` -> an equality with that cell, `Constant(c)` -> an equality with the
fixed column that holds c, `constrain_equal` / `assert_is_const` -> explicit equalities) and what `keygen_pk`
(src/scaffold/mod.rs:273) turns into the permutation argument.

The structure is data independent, so it is derived symbolically: `Sym` replays the gadgets' call tree with cell handles
instead of values, in exactly the cell order of halo2_vectordb_amd/csrc/gadgets.hpp and witness.hip (a copy arises wherever
the reference passes an assigned cell, by construction of the data flow here).  Small circuits are traced whole
(`trace_*`); the BASELINE-sized ones (hundreds of millions of cells) are assembled with numpy from traced unit blocks —
one distance, one per-vector assignment, one filter, one division — whose external inputs are resolved per instance
(`build_kmeans`, `build_nearest`).  tests/test_circuit_sym_cpu.py holds both against real witnesses: every gate flag,
every copy, every constant and every lookup source agree with the values of streams produced independently.

Cell codes of a traced block: src >= 0 copy of that block cell; SELF fresh witness; CONST constant (value in `cval`);
<= EXT0: external input number EXT0 - src.  [UPSTREAM-RECALL] for the halo2-base templates, as for the kernels.
"""
import math

import numpy as np

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
SELF, CONST, EXT0 = -1, -2, -10

EXP2_COEF = [3.6240421303547230336183979205877e-11, 4.1284327467833130245549169910389e-10, 0.0000000071086385644026346316624185550542,
             0.00000010172297085296590958930245291448, 0.0000013215904023658396206789543841996, 0.000015252713316417140696221389106544,
             0.00015403531076657894204857389177279, 0.0013333558131297097698435464957392, 0.0096181291078409107025643582456283,
             0.055504108664804181586140094858174, 0.24022650695910142332414229540187, 0.69314718055994529934452147700678, 1.0]      # fixed_point.rs:138-160
LOG_COEF = [-3.319586265362338e-08, 1.4957235315170112e-06, -3.1350053389526744e-05, 0.00040554177582512901, -0.0036218342998850703,
            0.023663846121538389, -0.11691877183255484, 0.44524062371564499, -1.3195777548208449, 3.0518128028712077, -5.4904626000399528,
            7.6298580090181591, -8.1653313719804235, 7.1389971101896279, -3.1937385492842112]                                      # fixed_point.rs:162-187
SIN_COEF = [-1.1008071636607462e-11, 2.4208013888629323e-10, -3.8584805817996712e-10, -2.3786993104309845e-08, -2.9795813710683115e-09,
            2.7608543130047009e-06, -6.4467066994122565e-09, -0.00019840680551418068, -3.839555844512214e-09, 0.0083333350601673614,
            -5.0943769725466814e-10, -0.16666666657583049, -8.5029878414113731e-12, 1.0000000000003146, -1.9323057584419828e-15]  # fixed_point.rs:189-211


class C:
    """QuantumCell::Constant"""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v % R


def ext(i):
    """handle of external input i of a block"""
    return EXT0 - i


def quantize(x, P):
    """fixed_point.rs:104-119 on a Python float: round half away from zero of |x| 2^P, negatives as r - q"""
    v = abs(x) * float(1 << P)
    q = int(math.floor(v))
    if v - math.floor(v) >= 0.5:
        q += 1
    return (R - q) % R if x < 0 else q


class Sym:
    """symbolic halo2-base Context: cell kinds instead of values"""

    def __init__(self, P, L):
        self.P, self.L = P, L
        self.src, self.gate, self.cval, self.asserted, self.lk = [], [], [], [], []

    def __len__(self):
        return len(self.src)

    # ------------------------------------------------------------------ Context
    def push(self, x, gate=False):
        p = len(self.src)
        if x is None:
            self.src.append(SELF)
            self.cval.append(None)
        elif isinstance(x, C):
            self.src.append(CONST)
            self.cval.append(x.v)
        else:
            assert x < p and (x >= 0 or x <= EXT0)
            self.src.append(x)
            self.cval.append(None)
        self.gate.append(bool(gate))
        return p

    def root(self, cell):
        while cell >= 0 and self.src[cell] >= 0:
            cell = self.src[cell]
        return cell

    def tie(self, a, fresh):
        """ctx.constrain_equal(a, fresh) where `fresh` is a witness cell of this block that copies nothing yet"""
        assert self.src[fresh] == SELF and a != fresh
        self.src[fresh] = a

    def tie_const(self, cell, value):
        """gate.assert_is_const(cell, value): the cell (through the cell it copies) is tied to the fixed column's `value`"""
        r = self.root(cell)
        assert r >= 0 and self.src[r] == SELF, "assert_is_const on a cell that is not rooted in this block"
        self.src[r] = CONST
        self.cval[r] = value % R
        self.asserted.append(r)

    def assign_witnesses(self, n):
        return [self.push(None) for _ in range(n)]

    def load_constant(self, v):
        return self.push(C(v))

    # ------------------------------------------------------------------ GateChip (flex_gate.rs)
    def g_add(self, a, b):
        self.push(a, True); self.push(b); self.push(C(1))
        return self.push(None)

    def g_sub(self, a, b):
        o = self.push(None, True)
        self.push(b); self.push(C(1)); self.push(a)
        return o

    def g_neg(self, a):
        self.push(a, True)
        o = self.push(None)
        self.push(C(1)); self.push(C(0))
        return o

    def g_mul(self, a, b):
        self.push(C(0), True); self.push(a); self.push(b)
        return self.push(None)

    def g_assert_bit(self, x):
        self.push(C(0), True); self.push(x); self.push(x); self.push(x)

    def g_not(self, a):
        return self.g_sub(C(1), a)

    def g_and(self, a, b):
        return self.g_mul(a, b)

    def g_or(self, a, b):
        nb = self.push(None, True)
        self.push(C(1)); self.push(b); self.push(C(1))
        self.push(b, True); self.push(a); self.push(nb)
        return self.push(None)

    def g_select(self, a, b, s):
        d = self.push(None, True)
        self.push(C(1)); self.push(b); self.push(a)
        self.push(b, True); self.push(s); self.push(d)
        return self.push(None)

    def g_is_zero(self, a):
        z = self.push(None, True)
        self.push(a); self.push(None); self.push(C(1))
        self.push(C(0), True); self.push(a)
        z2 = self.push(z)
        self.push(C(0))
        return z2                                            # ctx.get(-2)

    def g_is_equal(self, a, b):
        return self.g_is_zero(self.g_sub(a, b))

    def g_sum(self, cells):
        s = self.push(cells[0], len(cells) > 1)
        for i in range(1, len(cells)):
            self.push(cells[i]); self.push(C(1))
            s = self.push(None, i + 1 < len(cells))
        return s

    def g_select_by_indicator(self, cells, inds):
        n = len(cells)
        s = self.push(C(0), n > 0)
        for i in range(n):
            self.push(cells[i]); self.push(inds[i])
            s = self.push(None, i + 1 < n)
        return s

    def g_select_from_idx(self, cells, idx):
        inds = []
        for i in range(len(cells)):
            if i == 0:
                inds.append(self.g_is_zero(idx))             # the unrolled is_zero of idx_to_indicator
            else:
                d = self.push(None, True)                    # is_equal(idx, Constant(i)) = sub [d, i, 1, idx] + is_zero
                self.push(C(i)); self.push(C(1)); self.push(idx)
                inds.append(self.g_is_zero(d))
        return self.g_select_by_indicator(cells, inds)

    # ------------------------------------------------------------------ RangeChip (range.rs)
    def r_range_check(self, a, bits):
        L = self.L
        k, rem = -(-bits // L), bits % L
        if k == 1:
            self.lk.append(a)
            last = a
        else:
            limbs = [self.push(None, True)]
            acc = limbs[0]
            for i in range(1, k):
                limbs.append(self.push(None))
                self.push(C(1 << (i * L)))
                acc = self.push(None, i + 1 < k)
            self.tie(a, acc)                                 # ctx.constrain_equal(&a, &acc)
            self.lk.extend(limbs)
            last = limbs[-1]
        if rem == 1:
            self.g_assert_bit(last)
        elif rem > 1:
            last = self.g_mul(last, C(1 << (L - rem)))
            self.lk.append(last)
        return last

    def r_check_less_than(self, a, b, bits):
        chk = self.push(None, True)
        self.push(b); self.push(C(1))
        self.push(None, True)
        self.push(C(-(1 << bits))); self.push(C(1)); self.push(a)
        self.r_range_check(chk, bits)

    def r_check_big_less_than_safe(self, a, bound):
        rb = -(-bound.bit_length() // self.L) * self.L
        self.r_range_check(a, rb)
        self.r_check_less_than(a, C(bound), rb)

    def r_is_less_than(self, a, b, bits):
        padded = -(-bits // self.L) * self.L
        sh = self.push(None, True)
        self.push(b); self.push(C(1))
        self.push(None, True)
        self.push(C(-(1 << padded))); self.push(C(1)); self.push(a)
        return self.g_is_zero(self.r_range_check(sh, padded + self.L))

    def r_div_mod(self, a, shift, a_bits):
        """div_mod(a, 2^shift, a_bits) -> (div, rem)"""
        rem = self.push(None, True)
        self.push(C(1 << shift))
        div = self.push(None)
        self.push(a)
        self.r_check_big_less_than_safe(div, (1 << (a_bits - shift)) + 1)
        self.r_check_big_less_than_safe(rem, 1 << shift)
        return div, rem

    def r_div_mod_var(self, a, b, a_bits, b_bits):
        rem = self.push(None, True)
        self.push(b)
        div = self.push(None)
        self.push(a)
        self.r_range_check(div, a_bits)
        self.r_check_less_than(rem, b, b_bits)
        return div, rem

    # ------------------------------------------------------------------ FixedPointChip (fixed_point.rs)
    def is_neg(self, a):                                     # :523-539
        div, _ = self.r_div_mod(a, 2 * self.P + 1, 254)
        return self.g_not(self.g_is_zero(div))

    def qabs(self, a):                                       # :511-521
        rev = self.g_neg(a)
        n = self.is_neg(a)
        return self.g_select(rev, a, n)

    def cond_neg(self, a, flag):                             # :541-556
        return self.g_select(self.g_neg(a), a, flag)

    def signed_div_scale(self, a):                           # :974-1016
        P = self.P
        rem = self.push(None, True)
        self.push(C(1 << P))
        div = self.push(None)
        self.push(a)
        self.r_check_big_less_than_safe(rem, 1 << P)
        self.r_check_big_less_than_safe(self.qabs(div), 1 << (3 * P))
        return div

    def qmul(self, a, b):                                    # :588-604
        return self.signed_div_scale(self.g_mul(a, b))

    def bit_xor(self, a, b):                                 # :797-815
        a2 = self.g_add(C(0), a)
        b2 = self.g_add(C(0), b)
        self.g_assert_bit(a2)
        self.g_assert_bit(b2)
        ab = self.g_add(a2, b2)
        one = self.g_add(C(1), C(0))
        return self.g_is_equal(ab, one)

    def qdiv(self, a, b):                                    # :631-656
        P = self.P
        sa, sb = self.is_neg(a), self.is_neg(b)
        aa, ba = self.qabs(a), self.qabs(b)
        ar = self.g_mul(aa, C(1 << P))
        q, _ = self.r_div_mod_var(ar, ba, 4 * P, 2 * P)
        return self.cond_neg(q, self.bit_xor(sa, sb))

    def qmin(self, a, b):                                    # :936-952
        return self.g_select(a, b, self.is_neg(self.g_sub(a, b)))

    def polynomial(self, x, coef):                           # :658-686
        self.g_add(x, C(0))                                  # the dead qadd(x, 0)
        last, result = C(0), None
        for i, c in enumerate(coef):
            y = self.g_add(last, C(c))
            if i + 1 < len(coef):
                last = self.qmul(x, y)
            else:
                result = y
        return result

    def check_power_of_two(self, p2, e):                     # :688-708
        nb = 2 * self.P
        bits = [self.push(None, True)]                       # num_to_bits: inner_product(bits, pow_of_two), then assert_bit each
        acc = bits[0]
        for i in range(1, nb):
            bits.append(self.push(None))
            self.push(C(1 << i))
            acc = self.push(None, i + 1 < nb)
        self.tie(p2, acc)
        for b in bits:
            self.g_assert_bit(b)
        s = self.g_sum(bits)
        self.tie_const(self.g_is_zero(self.g_sub(s, C(1))), 1)
        bit = self.g_select_from_idx(bits, e)
        self.tie_const(self.g_is_zero(self.g_sub(bit, C(1))), 1)

    def qlog2(self, a):                                      # :736-795
        P = self.P
        a_assigned = self.g_add(a, C(0))
        is_neg = self.is_neg(a)
        is_zero = self.g_is_zero(a_assigned)
        self.tie_const(self.g_or(is_neg, is_zero), 0)
        pow1 = self.g_add(None, C(0))
        exp1 = self.g_add(None, C(0))
        self.check_power_of_two(pow1, exp1)
        pow2 = self.g_mul(pow1, C(2))
        exp2 = self.g_add(exp1, C(1))
        self.check_power_of_two(pow2, exp2)
        lt2 = self.r_is_less_than(a, pow2, 2 * P)
        gt1 = self.r_is_less_than(pow1, a, 2 * P)
        eq1 = self.g_is_equal(a, pow1)
        self.tie_const(self.g_and(lt2, self.g_or(eq1, gt1)), 1)
        shift = self.g_sub(C(P + 2), exp2)
        shift_neg = self.is_neg(shift)
        shift_abs = self.qabs(shift)
        spw = self.g_add(None, C(0))
        self.check_power_of_two(spw, shift_abs)
        a_ls = self.g_mul(a, spw)
        a_rs, _ = self.r_div_mod_var(a, spw, 2 * P, P + 1)
        a_norm = self.g_select(a_rs, a_ls, shift_neg)
        log_norm = self.polynomial(a_norm, [quantize(c, P) for c in LOG_COEF])
        lsq = self.g_mul(self.g_neg(shift), C(1 << P))
        return self.g_add(log_norm, lsq)

    def qexp2(self, a):                                      # :710-734
        P = self.P
        a_abs = self.qabs(a)
        ip, fpart = self.r_div_mod(a_abs, P, 2 * P)
        ip2 = self.g_select_from_idx([C(1 << i) for i in range(254)], ip)
        yf = self.polynomial(fpart, [quantize(c, P) for c in EXP2_COEF])
        res_pos = self.g_mul(ip2, yf)
        res_neg = self.qdiv(C(1 << P), res_pos)
        return self.g_select(res_neg, res_pos, self.is_neg(a))

    def qlog(self, a):                                       # :954-964
        l2e = self.load_constant(quantize(1.44269504088896340735992468100189214, self.P))
        return self.qdiv(self.qlog2(a), l2e)

    def qexp(self, a):                                       # :876-886
        ln2 = self.load_constant(quantize(0.693147180559945309417232121458176568, self.P))
        return self.qexp2(self.qdiv(a, ln2))

    def qsqrt(self, x):                                      # :966-972, 441-456
        half = self.load_constant(quantize(0.5, self.P))
        return self.qexp(self.qmul(half, self.qlog(x)))

    # the rest of FixedPointInstructions that examples/fixed_point.rs reaches
    def qmod(self, a, b):                                    # :606-629 (b: a positive constant or cell)
        P = self.P
        sa, sb = self.is_neg(a), self.is_neg(b)
        self.tie_const(sb, 0)                                # gate().assert_is_const(b_sign, 0)
        aa = self.qabs(a)
        _, r = self.r_div_mod_var(aa, b, 4 * P, 2 * P)
        return self.g_select(self.g_sub(b, r), r, sa)

    def qsin(self, a):                                       # :817-841
        P = self.P
        a_abs, a_sign = self.qabs(a), self.is_neg(a)
        a_mod = self.qmod(a_abs, C(quantize(math.pi * 2.0, P)))
        a_mpi = self.g_sub(a_mod, C(quantize(math.pi, P)))
        lower = self.is_neg(a_mpi)
        coef = [quantize(c, P) for c in SIN_COEF]
        s_mod = self.polynomial(a_mod, coef)
        s_mpi = self.g_neg(self.polynomial(a_mpi, coef))
        return self.cond_neg(self.g_select(s_mod, s_mpi, lower), a_sign)

    def sign(self, a):                                       # :558-569
        neg_one = self.g_neg(C(1))
        return self.g_select(neg_one, C(1), self.is_neg(a))

    def clip(self, a):                                       # :571-586
        sgn, aa = self.is_neg(a), self.qabs(a)
        _, rem = self.r_div_mod(aa, 2 * self.P, 254)
        return self.cond_neg(rem, sgn)

    def qcos(self, a):                                       # :843-852
        hp = self.load_constant(quantize(1.57079632679489661923132169163975144, self.P))
        return self.qsin(self.g_add(a, hp))

    def qtan(self, a):                                       # :383-393
        s = self.qsin(a)
        return self.qdiv(s, self.qcos(a))

    def _sinh_cosh(self, a, cosh):                           # :888-916
        ea = self.qexp(a)
        ena = self.qexp(self.g_neg(a))
        nume = self.g_add(ea, ena) if cosh else self.g_sub(ea, ena)
        return self.qdiv(nume, self.load_constant(quantize(2.0, self.P)))

    def qtanh(self, a):                                      # :407-417
        s = self._sinh_cosh(a, False)
        return self.qdiv(s, self._sinh_cosh(a, True))

    def fp_op(self, name, a):
        """one unary FixedPointInstructions call (the numbering of vdb_wit_fp_op names the same set)"""
        return dict(qexp2=self.qexp2, qlog2=self.qlog2, qsin=self.qsin, qexp=self.qexp, qlog=self.qlog, qsqrt=self.qsqrt, qabs=self.qabs,
                    is_neg=self.is_neg, neg=self.g_neg, signed_div_scale=self.signed_div_scale, sign=self.sign, clip=self.clip, qcos=self.qcos,
                    qtan=self.qtan, qsinh=lambda x: self._sinh_cosh(x, False), qcosh=lambda x: self._sinh_cosh(x, True), qtanh=self.qtanh)[name](a)

    def inner_product(self, a, b):                           # :854-874
        res = self.g_add(C(0), C(0))
        for x, y in zip(a, b):
            res = self.g_add(res, self.qmul(x, y))
        return res

    # ------------------------------------------------------------------ DistanceChip (distance.rs)
    def distance(self, metric, a, b):
        if metric == "euclidean":                            # :97-119
            ab = [self.g_sub(x, y) for x, y in zip(a, b)]
            return self.qsqrt(self.inner_product(ab, ab))
        if metric == "cosine":                               # :121-144
            ab, aa, bb = self.inner_product(a, b), self.inner_product(a, a), self.inner_product(b, b)
            sa, sb = self.qsqrt(aa), self.qsqrt(bb)
            sim = self.qdiv(ab, self.qmul(sa, sb))
            return self.g_sub(self.load_constant(quantize(1.0, self.P)), sim)
        if metric == "manhattan":                            # :177-195
            d = [self.g_sub(x, y) for x, y in zip(a, b)]
            return self.g_sum([self.qabs(x) for x in d])
        if metric == "hamming":                              # :146-175; `len` and `ab_sum_q` are load_witness cells nothing ties
            s = self.g_sum([self.g_is_equal(x, y) for x, y in zip(a, b)])
            len_q, sum_q = self.push(None), self.push(None)
            sim = self.qdiv(sum_q, len_q)
            return self.g_sub(self.load_constant(quantize(1.0, self.P)), sim)
        raise ValueError(metric)

    # ------------------------------------------------------------------ VectorDBChip (vectordb.rs)
    def nearest_vector(self, metric, query, vectors):        # :122-163, distance(v, query)
        dist = [self.distance(metric, v, query) for v in vectors]
        m = dist[0]
        for d in dist[1:]:
            m = self.qmin(m, d)
        ind = [self.g_is_equal(m, d) for d in dist]
        res = [self.g_select_by_indicator([v[j] for v in vectors], ind) for j in range(len(query))]
        return ind, res

    def kmeans(self, metric, vectors, K, I):                 # :225-362, distance(c, v)
        one = self.load_constant(quantize(1.0, self.P))
        zero = self.load_constant(0)
        cent = [list(v) for v in vectors[:K]]
        inds = None
        for _ in range(I):
            inds = []
            for v in vectors:
                inds.append(self.assign_block(one, zero, [self.distance(metric, c, v) for c in cent]))
            sizes = list(inds[0])
            for iv in inds[1:]:
                sizes = [self.g_add(s, x) for s, x in zip(sizes, iv)]
            for k in range(K):
                filt = [self.filter_block(zero, iv[k], v) for v, iv in zip(vectors, inds)]
                sums = list(filt[0])
                for f in filt[1:]:
                    sums = [self.g_add(x, s) for x, s in zip(f, sums)]                       # qadd(vector element, running sum)
                cent[k] = [self.qdiv(s, sizes[k]) for s in sums]
        return cent, inds

    def assign_block(self, one, zero, dist):
        """per vector: running qmin over its K distances, then (is_equal, select(one, zero, eq)) per cluster (:270-283)"""
        m = dist[0]
        for d in dist[1:]:
            m = self.qmin(m, d)
        return [self.g_select(one, zero, self.g_is_equal(m, d)) for d in dist]

    def filter_block(self, zero, sel, v):
        """is_zero(sel) and select(zero, v_j, is_zero) per dimension (:329-335)"""
        iz = self.g_is_zero(sel)
        return [self.g_select(zero, x, iz) for x in v]

    # ------------------------------------------------------------------ read-out
    def arrays(self):
        """(src, gate, const value index, constants, asserted cells, lookup sources) as numpy arrays of this block"""
        consts, cidx = {}, np.full(len(self.src), -1, dtype=np.int64)
        for p, v in enumerate(self.cval):
            if v is not None:
                cidx[p] = consts.setdefault(v, len(consts))
        return (np.asarray(self.src, dtype=np.int64), np.asarray(self.gate, dtype=bool), cidx, list(consts), np.asarray(self.asserted, dtype=np.int64),
                np.asarray(self.lk, dtype=np.int64))


# ---------------------------------------------------------------------------------------------------------------------------
# whole small circuits, cell by cell (the tests' ground truth for the block builders below)
class CopyMap:
    """copy constraints of a whole circuit over its flat advice stream:
    copy_of[i]    the earlier cell that cell i copies (i itself: none)
    const_idx[i]  index into `consts` (canonical integers) of the fixed-column value cell i is tied to, -1: none
    asserted[i]   the tie is an assert_is_const (the cell is a witness the circuit forces to that constant), not a Constant cell
    gate[i]       gate start
    lookup_src[j] the advice cell lookup cell j copies"""

    def __init__(self, copy_of, const_idx, consts, asserted, gate, lookup_src):
        self.copy_of, self.const_idx, self.consts, self.asserted, self.gate, self.lookup_src = copy_of, const_idx, consts, asserted, gate, lookup_src

    @property
    def n_cells(self):
        return self.copy_of.shape[0]

    def check_witness(self, values, lookup_values, flags=None):
        """every constraint of the map on a witness given as canonical Python / numpy integers (object arrays): returns a dict of
        violation counts.  `flags`: the kernels' flag bytes (bit 0 gate start, bit 1 constant): their gate bits must equal the
        map's and their constant bits must be a subset of the map's constants."""
        values = np.asarray(values, dtype=object)
        out = {}
        tied = np.flatnonzero(self.copy_of != np.arange(self.n_cells))
        out["copies_unequal"] = int(np.count_nonzero(values[tied] != values[self.copy_of[tied]]))
        cst = np.flatnonzero((self.const_idx >= 0) & ~self.asserted)
        out["constants_wrong"] = int(np.count_nonzero(values[cst] != np.asarray(self.consts, dtype=object)[self.const_idx[cst]]))
        asr = np.flatnonzero(self.asserted)
        out["asserts_violated"] = int(np.count_nonzero(values[asr] != np.asarray(self.consts, dtype=object)[self.const_idx[asr]]))
        lk = np.asarray(lookup_values, dtype=object)
        out["lookup_count"] = int(len(lk) != len(self.lookup_src))
        if len(lk) == len(self.lookup_src):
            out["lookup_copies_unequal"] = int(np.count_nonzero(lk != values[self.lookup_src])) if len(lk) else 0
        if flags is not None:
            flags = np.asarray(flags, dtype=np.uint8)
            out["gate_flags_differ"] = int(np.count_nonzero((flags & 1).astype(bool) != self.gate))
            out["kernel_constants_not_in_map"] = int(np.count_nonzero(((flags & 2) != 0) & ~((self.const_idx >= 0) & ~self.asserted)))
        return out


def _whole(sym, n_inputs):
    src, gate, cidx, consts, asserted, lk = sym.arrays()
    assert src.min() >= CONST, "external inputs in a whole circuit"
    n = len(src)
    copy_of = np.where(src >= 0, src, np.arange(n))
    am = np.zeros(n, dtype=bool)
    am[asserted] = True
    return CopyMap(copy_of, cidx, consts, am, gate, lk)


def trace_distance(metric, dim, P, L, n_pairs=1):
    """ctx.assign_witnesses(a); ctx.assign_witnesses(b); distance(a, b) — n_pairs times on the same two vectors (examples/euclid.rs)"""
    s = Sym(P, L)
    a, b = s.assign_witnesses(dim), s.assign_witnesses(dim)
    outs = [s.distance(metric, a, b) for _ in range(n_pairs)]
    return _whole(s, 2 * dim), outs


def trace_distances(metrics, dim, P, L):
    """ctx.assign_witnesses(a); ctx.assign_witnesses(b); then one distance after the other on the same two vectors
    (examples/distances.rs:40-59; pipeline.DistancesHotPath)"""
    s = Sym(P, L)
    a, b = s.assign_witnesses(dim), s.assign_witnesses(dim)
    outs = [s.distance(m, a, b) for m in metrics]
    return _whole(s, 2 * dim), outs


def trace_fixed_point(ops, P, L):
    """examples/fixed_point.rs:55-111: x = ctx.load_witness(..), then one FixedPointInstructions call after the other on x; -> the map and the
    cells the example makes public: x, then every result"""
    s = Sym(P, L)
    (x,) = s.assign_witnesses(1)
    outs = [x] + [s.fp_op(name, x) for name in ops]
    return _whole(s, 1), outs


def trace_nearest(metric, n, dim, P, L):
    s = Sym(P, L)
    q = s.assign_witnesses(dim)
    vs = [s.assign_witnesses(dim) for _ in range(n)]
    ind, res = s.nearest_vector(metric, q, vs)
    return _whole(s, (n + 1) * dim), (ind, res)


def trace_kmeans(metric, n, dim, K, I, P, L):
    s = Sym(P, L)
    vs = [s.assign_witnesses(dim) for _ in range(n)]
    cent, inds = s.kmeans(metric, vs, K, I)
    return _whole(s, n * dim), (cent, inds)


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE-sized circuits: traced unit blocks, instantiated with numpy
class Block:
    """a traced unit with `n_ext` external inputs and `outs` (block cells handed to later blocks)"""

    def __init__(self, sym, outs):
        self.src, self.gate, self.cidx, self.consts, self.asserted, self.lk = sym.arrays()
        self.n, self.n_lk, self.outs = len(self.src), len(self.lk), np.asarray(outs, dtype=np.int64)
        self.local = self.src >= 0
        self.isext = self.src <= EXT0
        self.ext_no = np.where(self.isext, EXT0 - self.src, 0)
        self.lk_isext = self.lk <= EXT0
        self.lk_ext_no = np.where(self.lk_isext, EXT0 - self.lk, 0)


class _Builder:
    def __init__(self, n_cells, n_lookup):
        self.copy_of = np.arange(n_cells, dtype=np.int64)
        self.const_idx = np.full(n_cells, -1, dtype=np.int64)
        self.asserted = np.zeros(n_cells, dtype=bool)
        self.gate = np.zeros(n_cells, dtype=bool)
        self.lookup_src = np.full(n_lookup, -1, dtype=np.int64)
        self.consts, self._cmap = [], {}

    def const_index(self, v):
        if v not in self._cmap:
            self._cmap[v] = len(self.consts)
            self.consts.append(v)
        return self._cmap[v]

    def constant_cell(self, pos, value):
        self.const_idx[pos] = self.const_index(value % R)

    def place(self, blk, bases, lk_bases, ext_cells):
        """instances of `blk` at advice offsets `bases` (m,), lookup offsets `lk_bases` (m,), external inputs `ext_cells` (m, n_ext);
        returns the absolute cells of the block's outputs, (m, n_outs)"""
        bases = np.asarray(bases, dtype=np.int64).reshape(-1)
        m = bases.size
        ext_cells = np.asarray(ext_cells, dtype=np.int64).reshape(m, -1)
        remap = np.asarray([self.const_index(v) for v in blk.consts], dtype=np.int64)
        cid = np.where(blk.cidx >= 0, remap[np.maximum(blk.cidx, 0)] if len(remap) else -1, -1)
        step = max(1, (1 << 24) // max(blk.n, 1))                      # bounded temporaries
        for lo in range(0, m, step):
            b = bases[lo: lo + step, None]
            idx = b + np.arange(blk.n, dtype=np.int64)[None, :]
            val = np.where(blk.local[None, :], b + np.maximum(blk.src, 0)[None, :], idx)
            if blk.isext.any():
                val = np.where(blk.isext[None, :], ext_cells[lo: lo + step][:, blk.ext_no], val)
            self.copy_of[idx.reshape(-1)] = val.reshape(-1)
            self.const_idx[idx.reshape(-1)] = np.broadcast_to(cid[None, :], idx.shape).reshape(-1)
            self.gate[idx.reshape(-1)] = np.broadcast_to(blk.gate[None, :], idx.shape).reshape(-1)
            if blk.asserted.size:
                self.asserted[(b + blk.asserted[None, :]).reshape(-1)] = True
            if blk.n_lk:
                lb = np.asarray(lk_bases, dtype=np.int64).reshape(-1)[lo: lo + step, None]
                lval = np.where(blk.lk_isext[None, :], ext_cells[lo: lo + step][:, blk.lk_ext_no], b + np.maximum(blk.lk, 0)[None, :])
                self.lookup_src[(lb + np.arange(blk.n_lk, dtype=np.int64)[None, :]).reshape(-1)] = lval.reshape(-1)
        return bases[:, None] + blk.outs[None, :]

    def finish(self):
        assert (self.lookup_src >= 0).all(), "lookup cells without a source"
        return CopyMap(self.copy_of, self.const_idx, self.consts, self.asserted, self.gate, self.lookup_src)


def _distance_block(metric, dim, P, L):
    s = Sym(P, L)
    out = s.distance(metric, [ext(i) for i in range(dim)], [ext(dim + i) for i in range(dim)])
    return Block(s, [out])


def build_nearest(metric, n, dim, P, L, builder=None, extra_cells=0, finish=True):
    """nearest_vector(query, vectors) after [query | vectors] have been assigned (tests/vectordb/mod.rs:220-247): the map the
    whole-circuit trace gives, assembled from one distance block, one qmin block and the closing cells.  `extra_cells`: room for a
    gadget that follows in the same stream (the query circuit's merkle_commitment); with finish=False the builder itself is returned,
    (builder, outputs, cells used so far), for the caller to go on placing"""
    db = _distance_block(metric, dim, P, L)
    s = Sym(P, L)
    qm = Block(s, [s.qmin(ext(0), ext(1))])
    n_in = (n + 1) * dim
    dist0, qmin0 = n_in, n_in + n * db.n
    iseq0 = qmin0 + (n - 1) * qm.n
    sel0 = iseq0 + 12 * n
    total = sel0 + dim * (1 + 3 * n)
    B = (builder or _Builder)(total + extra_cells, n * db.n_lk + (n - 1) * qm.n_lk)
    query = np.arange(dim, dtype=np.int64)
    vec = dim + np.arange(n * dim, dtype=np.int64).reshape(n, dim)
    i = np.arange(n, dtype=np.int64)
    d = B.place(db, dist0 + i * db.n, i * db.n_lk, np.concatenate([vec, np.broadcast_to(query, (n, dim))], axis=1))[:, 0]
    acc = np.empty(n, dtype=np.int64)
    acc[0] = d[0]
    if n > 1:
        j = np.arange(n - 1, dtype=np.int64)
        outs = qmin0 + j * qm.n + qm.outs[0]
        acc[1:] = outs
        B.place(qm, qmin0 + j * qm.n, n * db.n_lk + j * qm.n_lk, np.stack([acc[:-1], d[1:]], axis=1))
    s = Sym(P, L)
    ie = Block(s, [s.g_is_equal(ext(0), ext(1))])
    ind = B.place(ie, iseq0 + 12 * i, np.zeros(n, dtype=np.int64), np.stack([np.full(n, acc[-1]), d], axis=1))[:, 0]
    s = Sym(P, L)
    sb = Block(s, [s.g_select_by_indicator([ext(k) for k in range(n)], [ext(n + k) for k in range(n)])])
    jd = np.arange(dim, dtype=np.int64)
    res = B.place(sb, sel0 + jd * (1 + 3 * n), np.zeros(dim, dtype=np.int64), np.concatenate([vec.T, np.broadcast_to(ind, (dim, n))], axis=1))[:, 0]
    if not finish:
        return B, (ind, res), total
    return B.finish(), (ind, res)


def build_kmeans(metric, n, dim, K, I, P, L, builder=None):
    """kmeans::<K, I>(vectors) after the vectors have been assigned (examples/kmeans.rs:40-49), in the stream order of
    witness.hip (KmLayout): [one, zero] then per iteration N x (K distances, assignment), the sizes chain, and per cluster
    (N filters, the sums chain, D divisions).  `builder`: the class that holds the map's arrays (default: numpy on the host;
    circuit_dev.DeviceBuilder assembles them on the GPU)"""
    db = _distance_block(metric, dim, P, L)
    s = Sym(P, L)
    ab = Block(s, s.assign_block(ext(0), ext(1), [ext(2 + k) for k in range(K)]))
    s = Sym(P, L)
    fb = Block(s, s.filter_block(ext(0), ext(1), [ext(2 + j) for j in range(dim)]))
    s = Sym(P, L)
    qd = Block(s, [s.qdiv(ext(0), ext(1))])
    s = Sym(P, L)
    add = Block(s, [s.g_add(ext(0), ext(1))])
    per_vec, per_vec_l = K * db.n + ab.n, K * db.n_lk + ab.n_lk
    assign, assign_l = n * per_vec, n * per_vec_l
    sizes = (n - 1) * K * 4
    per_cluster, per_cluster_l = n * fb.n + (n - 1) * dim * 4 + dim * qd.n, dim * qd.n_lk
    it_cells, it_lk = assign + sizes + K * per_cluster, assign_l + K * per_cluster_l
    n_in = n * dim
    total, total_l = n_in + 2 + I * it_cells, I * it_lk
    B = (builder or _Builder)(total, total_l)
    one, zero = n_in, n_in + 1
    B.constant_cell(one, quantize(1.0, P))
    B.constant_cell(zero, 0)
    vec = np.arange(n * dim, dtype=np.int64).reshape(n, dim)
    cent = vec[:K].copy()
    v = np.arange(n, dtype=np.int64)
    ind = None
    for it in range(I):
        pos, lpos = n_in + 2 + it * it_cells, it * it_lk
        vk = np.repeat(v, K)
        kk = np.tile(np.arange(K, dtype=np.int64), n)
        d = B.place(db, pos + vk * per_vec + kk * db.n, lpos + vk * per_vec_l + kk * db.n_lk, np.concatenate([cent[kk], vec[vk]], axis=1))[:, 0].reshape(n, K)
        ind = B.place(ab, pos + v * per_vec + K * db.n, lpos + v * per_vec_l + K * db.n_lk,
                      np.concatenate([np.full((n, 1), one), np.full((n, 1), zero), d], axis=1))          # (n, K)
        # cluster sizes: for v = 1 .. n-1, for k: qadd(size_k, ind[v][k]); the fold starts from ind[0]
        sb = pos + assign
        size_cells = ind[0].copy()
        if n > 1:
            vv = np.repeat(np.arange(1, n, dtype=np.int64), K)
            kk2 = np.tile(np.arange(K, dtype=np.int64), n - 1)
            bases = sb + (vv - 1) * 4 * K + 4 * kk2
            outs = bases + add.outs[0]
            prev = np.where(vv == 1, ind[0][kk2], outs - 4 * K)
            B.place(add, bases, np.zeros_like(bases), np.stack([prev, ind[vv, kk2]], axis=1))
            size_cells = outs.reshape(n - 1, K)[-1]
        new_cent = np.empty((K, dim), dtype=np.int64)
        for k in range(K):
            cb, clb = pos + assign + sizes + k * per_cluster, lpos + assign_l + k * per_cluster_l
            filt = B.place(fb, cb + v * fb.n, np.zeros(n, dtype=np.int64), np.concatenate([np.full((n, 1), zero), ind[:, k: k + 1], vec], axis=1))    # (n, dim)
            sums0 = cb + n * fb.n
            sum_cells = filt[0].copy()
            if n > 1:
                vv = np.repeat(np.arange(1, n, dtype=np.int64), dim)
                jj = np.tile(np.arange(dim, dtype=np.int64), n - 1)
                bases = sums0 + (vv - 1) * 4 * dim + 4 * jj
                outs = bases + add.outs[0]
                prev = np.where(vv == 1, filt[0][jj], outs - 4 * dim)
                B.place(add, bases, np.zeros_like(bases), np.stack([filt[vv, jj], prev], axis=1))    # qadd(vector element, running sum)
                sum_cells = outs.reshape(n - 1, dim)[-1]
            j = np.arange(dim, dtype=np.int64)
            new_cent[k] = B.place(qd, sums0 + (n - 1) * dim * 4 + j * qd.n, clb + j * qd.n_lk, np.stack([sum_cells, np.full(dim, size_cells[k])], axis=1))[:, 0]
        cent = new_cent
    return B.finish(), (cent, ind)
