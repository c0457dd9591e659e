"""ORACLE — TEST INFRASTRUCTURE ONLY.

Second, independently written restatement (pure Python big ints, value level only — no cell
streams) of the same reference arithmetic, used to cross-check oracle/liboracle.so so that two
implementations written from the reference's source agree.  Small cases only (it is slow).

Follows /root/reference/src/gadget/fixed_point.rs, distance.rs, vectordb.rs (lines cited inline)
and the public Poseidon parameter-generation procedure.  PARITY UNPINNED vs the Rust binary.
"""
import math
import struct

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


# ------------------------------------------------------------------ fixed point (values in Z_r)
class FixedPoint:
    def __init__(self, P=48):
        self.P = P
        self.scale = 1 << P
        self.negative_point = R - (1 << (2 * P + 1))  # fixed_point.rs:76
        self.exp2_poly = [self.quantize(c) for c in (
            3.6240421303547230336183979205877e-11, 4.1284327467833130245549169910389e-10,
            0.0000000071086385644026346316624185550542, 0.00000010172297085296590958930245291448,
            0.0000013215904023658396206789543841996, 0.000015252713316417140696221389106544,
            0.00015403531076657894204857389177279, 0.0013333558131297097698435464957392,
            0.0096181291078409107025643582456283, 0.055504108664804181586140094858174,
            0.24022650695910142332414229540187, 0.69314718055994529934452147700678, 1.0)]
        self.log_poly = [self.quantize(c) for c in (
            -3.319586265362338e-08, 1.4957235315170112e-06, -3.1350053389526744e-05, 0.00040554177582512901,
            -0.0036218342998850703, 0.023663846121538389, -0.11691877183255484, 0.44524062371564499,
            -1.3195777548208449, 3.0518128028712077, -5.4904626000399528, 7.6298580090181591,
            -8.1653313719804235, 7.1389971101896279, -3.1937385492842112)]

    # fixed_point.rs:104-119
    def quantize(self, x):
        neg = (not math.isnan(x)) and math.copysign(1.0, x) < 0
        y = abs(x) * float(self.scale)
        if math.isnan(y):
            q = 0
        elif math.isinf(y):
            q = (1 << 128) - 1
        else:
            # f64::round = half away from zero
            fl = math.floor(y)
            q = int(fl) + (1 if y - fl >= 0.5 else 0)
            q = min(q, (1 << 128) - 1)
        return (R - q) % R if neg else q

    # fixed_point.rs:121-136
    def dequantize(self, x):
        sign = 1.0
        if x > self.negative_point:
            x = (R - 1 - x - 1) % R
            sign = -1.0
        lo = x & ((1 << 128) - 1)
        return sign * (float(lo // self.scale) + float(lo % self.scale) / float(self.scale))

    def is_neg(self, a):  # :523-539  quotient of a / 2^(2P+1) non-zero
        return 1 if (a >> (2 * self.P + 1)) != 0 else 0

    def qabs(self, a):  # :511-521
        return (R - a) % R if self.is_neg(a) else a

    def signed_div_scale(self, a):  # :974-1016
        if a > (1 << 252):
            a_abs = R - a
            q = R - (-(-a_abs // self.scale))
            r = a - (self.scale * q) % R
            assert 0 <= r < self.scale
            return q % R, r
        return a >> self.P, a & (self.scale - 1)

    def qmul(self, a, b):  # :588-604
        return self.signed_div_scale(a * b % R)[0]

    def qdiv(self, a, b):  # :631-656
        sa, sb = self.is_neg(a), self.is_neg(b)
        aa, ba = self.qabs(a), self.qabs(b)
        q = (aa * self.scale % R) // ba
        return (R - q) % R if sa ^ sb else q

    def polynomial(self, x, coef):  # :658-686
        y = 0
        for i, c in enumerate(coef):
            y = (y + c) % R
            if i < len(coef) - 1:
                y = self.qmul(x, y)
        return y

    def qlog2(self, a):  # :736-795
        nd = a.bit_length() - 1 if a else 1
        exp2 = nd + 1
        shift = (self.P + 2 - exp2) % R
        shift_neg = self.is_neg(shift)
        shift_abs = self.qabs(shift)
        sp = 1 << (shift_abs & 0xFFFFFFFF)
        a_norm = (a // sp) if shift_neg else (a * sp % R)
        log_norm = self.polynomial(a_norm, self.log_poly)
        return (log_norm + (R - shift) % R * self.scale) % R

    def qexp2(self, a):  # :710-734
        a_abs = self.qabs(a)
        ip, fp = a_abs >> self.P, a_abs & (self.scale - 1)
        assert ip < 254
        res_pos = (1 << ip) % R * self.polynomial(fp, self.exp2_poly) % R
        return self.qdiv(self.scale, res_pos) if self.is_neg(a) else res_pos

    def qlog(self, a):  # :954-964
        return self.qdiv(self.qlog2(a), self.quantize(math.log2(math.e)))

    def qexp(self, a):  # :876-886
        return self.qexp2(self.qdiv(a, self.quantize(math.log(2.0))))

    def qpow(self, x, e):  # :441-456
        return self.qexp(self.qmul(e, self.qlog(x)))

    def qsqrt(self, x):  # :966-972
        return self.qpow(x, self.quantize(0.5))

    def qmin(self, a, b):  # :936-952
        return a if self.is_neg((a - b) % R) else b

    def qmax(self, a, b):  # :918-934
        return b if self.is_neg((a - b) % R) else a

    # the rest of FixedPointInstructions (examples/fixed_point.rs reaches qsin)
    SIN_POLY = (-1.1008071636607462e-11, 2.4208013888629323e-10, -3.8584805817996712e-10, -2.3786993104309845e-08, -2.9795813710683115e-09,
                2.7608543130047009e-06, -6.4467066994122565e-09, -0.00019840680551418068, -3.839555844512214e-09, 0.0083333350601673614,
                -5.0943769725466814e-10, -0.16666666657583049, -8.5029878414113731e-12, 1.0000000000003146, -1.9323057584419828e-15)

    def sign(self, a):  # :558-569: the field elements 1 / -1, not quantized ones
        return R - 1 if self.is_neg(a) else 1

    def clip(self, a):  # :571-586: |a| mod 2^(2P), the sign restored
        m = self.qabs(a) % (1 << (2 * self.P))
        return (R - m) % R if self.is_neg(a) else m

    def qmod(self, a, b):  # :606-629, b positive
        r = self.qabs(a) % b
        return (b - r) % R if self.is_neg(a) else r

    def qsin(self, a):  # :817-841
        coef = [self.quantize(c) for c in self.SIN_POLY]
        a_mod = self.qmod(self.qabs(a), self.quantize(math.pi * 2.0))
        a_mpi = (a_mod - self.quantize(math.pi)) % R
        s = self.polynomial(a_mod, coef) if self.is_neg(a_mpi) else (R - self.polynomial(a_mpi, coef)) % R
        return (R - s) % R if self.is_neg(a) else s

    def qcos(self, a):  # :843-852
        return self.qsin((a + self.quantize(math.pi / 2)) % R)

    def qtan(self, a):  # :383-393
        return self.qdiv(self.qsin(a), self.qcos(a))

    def qsinh(self, a):  # :888-901
        return self.qdiv((self.qexp(a) - self.qexp((R - a) % R)) % R, self.quantize(2.0))

    def qcosh(self, a):  # :903-916
        return self.qdiv((self.qexp(a) + self.qexp((R - a) % R)) % R, self.quantize(2.0))

    def qtanh(self, a):  # :407-417
        return self.qdiv(self.qsinh(a), self.qcosh(a))

    def inner_product(self, a, b):  # :854-874
        res = 0
        for x, y in zip(a, b):
            res = (res + self.qmul(x, y)) % R
        return res

    # distance.rs
    def euclidean(self, a, b):  # :97-119
        d = [(x - y) % R for x, y in zip(a, b)]
        return self.qsqrt(self.inner_product(d, d))

    def cosine(self, a, b):  # :121-144
        ab, aa, bb = self.inner_product(a, b), self.inner_product(a, a), self.inner_product(b, b)
        den = self.qmul(self.qsqrt(aa), self.qsqrt(bb))
        return (self.quantize(1.0) - self.qdiv(ab, den)) % R

    def manhattan(self, a, b):  # :177-195
        return sum(self.qabs((x - y) % R) for x, y in zip(a, b)) % R

    def hamming(self, a, b):  # :146-175: one minus the share of equal elements; both counts quantized as f64 values
        same = sum(1 for x, y in zip(a, b) if x == y)
        return (self.quantize(1.0) - self.qdiv(self.quantize(float(same)), self.quantize(float(len(a))))) % R

    # vectordb.rs:122-163
    def nearest_vector(self, query, vectors, dist):
        d = [dist(v, query) for v in vectors]
        m = d[0]
        for x in d[1:]:
            m = self.qmin(m, x)
        ind = [1 if x == m else 0 for x in d]
        res = []
        for j in range(len(query)):
            s = 0
            for v, i in zip(vectors, ind):
                if i:
                    s = v[j]
            res.append(s)
        return ind, res

    # vectordb.rs:225-362
    def kmeans(self, vectors, K, I, dist):
        one = self.quantize(1.0)
        cent = [list(v) for v in vectors[:K]]
        inds = []
        for _ in range(I):
            inds = []
            for v in vectors:
                d = [dist(c, v) for c in cent]
                m = d[0]
                for x in d[1:]:
                    m = self.qmin(m, x)
                inds.append([one if x == m else 0 for x in d])
            sizes = [sum(i[k] for i in inds) % R for k in range(K)]
            new = []
            for k in range(K):
                s = [0] * len(vectors[0])
                for v, i in zip(vectors, inds):
                    if i[k] != 0:
                        s = [(x + y) % R for x, y in zip(s, v)]
                new.append([self.qdiv(x, sizes[k]) for x in s])
            cent = new
        return cent, inds


# ------------------------------------------------------------------ Poseidon (x^5, t=3, 8/57)
class Grain:
    def __init__(self, n, t, r_f, r_p):
        bits = []
        for v, w in ((1, 2), (0, 4), (n, 12), (t, 12), (r_f, 10), (r_p, 10)):
            bits += [(v >> i) & 1 for i in range(w - 1, -1, -1)]
        self.s = bits + [1] * 30
        for _ in range(160):
            self._upd()

    def _upd(self):
        s = self.s
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb

    def bit(self):
        b = self._upd()
        while b == 0:
            self._upd()
            b = self._upd()
        return self._upd()

    def integer(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v

    def field(self, n, p, reject=True):
        while True:
            v = self.integer(n)
            if not reject:
                return v % p
            if v < p:
                return v


class Poseidon:
    def __init__(self, t=3, r_f=8, r_p=57):
        g = Grain(254, t, r_f, r_p)
        self.t, self.r_f, self.r_p = t, r_f, r_p
        self.rc = [[g.field(254, R) for _ in range(t)] for _ in range(r_f + r_p)]
        xs = [g.field(254, R, False) for _ in range(t)]
        ys = [g.field(254, R, False) for _ in range(t)]
        self.mds = [[pow(xs[i] + ys[j], -1, R) for j in range(t)] for i in range(t)]

    def permute(self, st):
        st = list(st)
        half = self.r_f // 2
        for r in range(self.r_f + self.r_p):
            st = [(x + c) % R for x, c in zip(st, self.rc[r])]
            if r < half or r >= half + self.r_p:
                st = [pow(x, 5, R) for x in st]
            else:
                st[0] = pow(st[0], 5, R)
            st = [sum(self.mds[i][j] * st[j] for j in range(self.t)) % R for i in range(self.t)]
        return st

    def hash(self, msg):
        """Sponge as the halo2-lib PoseidonChip [UPSTREAM-RECALL]: state [2^64,0,0]; absorb RATE
        elements per permutation; a short chunk gets +1 after its last element; if the last chunk was
        full (or the message is empty) one extra permutation absorbs only the +1; output state[1]."""
        rate = self.t - 1
        st = [1 << 64] + [0] * rate
        chunks = [msg[i:i + rate] for i in range(0, len(msg), rate)]
        pad = 0
        for ch in chunks:
            pad = rate - len(ch)
            for i, v in enumerate(ch):
                st[1 + i] = (st[1 + i] + v) % R
            if len(ch) < rate:
                st[1 + len(ch)] = (st[1 + len(ch)] + 1) % R
            st = self.permute(st)
        if pad == 0:
            st[1] = (st[1] + 1) % R
            st = self.permute(st)
        return st[1]

    def merkle_root(self, vectors):  # vectordb.rs:165-223
        lv = [self.hash(v) for v in vectors]
        n = 1
        while n < len(lv):
            n <<= 1
        lv += [0] * (n - len(lv))
        while len(lv) > 1:
            lv = [self.hash([lv[i], lv[i + 1]]) for i in range(0, len(lv), 2)]
        return lv[0]


# ------------------------------------------------------------------ G1 (affine, python ints)
def g1_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return x3, (lam * (x1 - x3) - y1) % Q


def g1_mul(p, k):
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, p)
        p = g1_add(p, p)
        k >>= 1
    return acc


def msm(scalars, points):
    acc = None
    for s, p in zip(scalars, points):
        acc = g1_add(acc, g1_mul(p, s % R))
    return acc


def dft(a, omega):
    n = len(a)
    return [sum(a[j] * pow(omega, i * j, R) for j in range(n)) % R for i in range(n)]


def f64_bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]
