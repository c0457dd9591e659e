"""BN254 optimal ate pairing on Python integers — TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md §2: nothing under
halo2_vectordb_amd/ may import this).  It lets tests/test_gpu_rounds.py check openings the way a verifier does, with the
pairing equation and the G2 side of the SRS instead of the toxic scalar.

Public standard, restated from the definition (Barreto–Naehrig curve y^2 = x^3 + 3 over Fq, embedding degree 12, twist
y^2 = x^3 + 3 / (9 + u) over Fq2 = Fq[u] / (u^2 + 1), Fq12 = Fq[w] / (w^12 - 18 w^6 + 82), ate loop count 6x + 2 with
x = 4965661367192848881; G2 generator as in EIP-197).  Self-checks in tests/test_oracle_field.py: the generator lies on the
twist and has order r, the pairing is bilinear and non-degenerate."""

Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ATE_LOOP_COUNT = 29793968203157093288          # 6 x + 2
LOG_ATE = 63
FQ12_MOD = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]     # w^12 - 18 w^6 + 82
FQ2_MOD = [1, 0]                                        # u^2 + 1


class FQP:
    """element of Fq[X] / (X^d + sum mod[i] X^i)"""
    __slots__ = ("c", "mod")

    def __init__(self, coeffs, mod):
        self.c = [v % Q for v in coeffs]
        self.mod = mod

    def _wrap(self, o):
        if isinstance(o, FQP):
            return o
        return FQP([o] + [0] * (len(self.c) - 1), self.mod)

    def __add__(self, o):
        o = self._wrap(o)
        return FQP([a + b for a, b in zip(self.c, o.c)], self.mod)

    def __sub__(self, o):
        o = self._wrap(o)
        return FQP([a - b for a, b in zip(self.c, o.c)], self.mod)

    def __neg__(self):
        return FQP([-a for a in self.c], self.mod)

    def __mul__(self, o):
        if not isinstance(o, FQP):
            return FQP([a * o for a in self.c], self.mod)
        d = len(self.c)
        b = [0] * (2 * d - 1)
        for i, x in enumerate(self.c):
            if x:
                for j, y in enumerate(o.c):
                    b[i + j] += x * y
        for exp in range(2 * d - 2, d - 1, -1):
            top = b[exp] % Q
            if top:
                for i, m in enumerate(self.mod):
                    if m:
                        b[exp - d + i] -= top * m
            b[exp] = 0
        return FQP(b[:d], self.mod)

    __rmul__ = __mul__

    def __eq__(self, o):
        o = self._wrap(o)
        return self.c == o.c

    def is_zero(self):
        return not any(self.c)

    def __pow__(self, e):
        out = FQP([1] + [0] * (len(self.c) - 1), self.mod)
        base = self
        while e:
            if e & 1:
                out = out * base
            base = base * base
            e >>= 1
        return out

    def inv(self):
        """extended Euclid on polynomials over Fq"""
        d = len(self.c)
        lm, hm = [1] + [0] * d, [0] * (d + 1)
        low, high = self.c + [0], list(self.mod) + [1]
        deg = lambda p: max([i for i, v in enumerate(p) if v % Q] or [0])
        while deg(low):
            dl, dh = deg(low), deg(high)
            # r = high // low (polynomial division)
            r = [0] * (d + 1)
            tmp = [v % Q for v in high]
            inv_lead = pow(low[dl] % Q, -1, Q)
            for i in range(dh - dl, -1, -1):
                coef = tmp[dl + i] * inv_lead % Q
                r[i] = coef
                for j in range(dl + 1):
                    tmp[i + j] = (tmp[i + j] - coef * low[j]) % Q
            nm, new = [v % Q for v in hm], [v % Q for v in high]
            for i in range(d + 1):
                for j in range(d + 1 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % Q
                    new[i + j] = (new[i + j] - low[i] * r[j]) % Q
            lm, low, hm, high = nm, new, lm, low
        s = pow(low[0] % Q, -1, Q)
        return FQP([v * s for v in lm[:d]], self.mod)

    def __truediv__(self, o):
        return self * self._wrap(o).inv()


def fq2(a, b):
    return FQP([a, b], FQ2_MOD)


def fq12(coeffs):
    return FQP(coeffs, FQ12_MOD)


FQ12_ONE = fq12([1] + [0] * 11)
W = fq12([0, 1] + [0] * 10)
B2 = fq2(3, 0) / fq2(9, 1)                       # twist coefficient 3 / (9 + u)
G2 = (fq2(10857046999023057135944570762232829481370756359578518086990519993285655852781,
          11559732032986387107991004021392285783925812861821192530917403151452391805634),
      fq2(8495653923123431417604973247489272438418190587263600148770280649306958101930,
          4082367875863433681332203403145435568316851327593401208105741076214120093531))
G1 = (1, 2)


# ---- generic affine arithmetic on y^2 = x^3 + b over any of the fields above (None = the point at infinity)
def _is_fqp(v):
    return isinstance(v, FQP)


def _inv(v):
    return v.inv() if _is_fqp(v) else pow(v % Q, -1, Q)


def _eq(a, b):
    return (a == b) if _is_fqp(a) else (a - b) % Q == 0


def pt_double(p):
    if p is None:
        return None
    x, y = p
    lam = 3 * x * x * _inv(2 * y)
    nx = lam * lam - 2 * x
    ny = lam * (x - nx) - y
    return (nx, ny) if _is_fqp(nx) else (nx % Q, ny % Q)


def pt_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if _eq(x1, x2):
        if _eq(y1, y2):
            return pt_double(p)
        return None
    lam = (y2 - y1) * _inv(x2 - x1)
    nx = lam * lam - x1 - x2
    ny = lam * (x1 - nx) - y1
    return (nx, ny) if _is_fqp(nx) else (nx % Q, ny % Q)


def pt_mul(p, k):
    acc = None
    while k:
        if k & 1:
            acc = pt_add(acc, p)
        p = pt_double(p)
        k >>= 1
    return acc


def pt_neg(p):
    if p is None:
        return None
    return (p[0], -p[1]) if _is_fqp(p[1]) else (p[0], (-p[1]) % Q)


def on_twist(p):
    x, y = p
    return (y * y - x * x * x - B2).is_zero()


def twist(p):
    """G2 point over Fq2 -> the same point on y^2 = x^3 + 3 over Fq12"""
    if p is None:
        return None
    x, y = p
    xc = [x.c[0] - 9 * x.c[1], x.c[1]]
    yc = [y.c[0] - 9 * y.c[1], y.c[1]]
    nx = fq12([xc[0]] + [0] * 5 + [xc[1]] + [0] * 5)
    ny = fq12([yc[0]] + [0] * 5 + [yc[1]] + [0] * 5)
    return nx * W ** 2, ny * W ** 3


def cast_g1(p):
    x, y = p
    return fq12([x] + [0] * 11), fq12([y] + [0] * 11)


def linefunc(p1, p2, t):
    """the line through p1, p2 (tangent when equal) evaluated at t; all over Fq12"""
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if not (x1 == x2):
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = 3 * x1 * x1 / (2 * y1)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(q, p):
    if q is None or p is None:
        return FQ12_ONE
    r, f = q, FQ12_ONE
    for i in range(LOG_ATE, -1, -1):
        f = f * f * linefunc(r, r, p)
        r = pt_double(r)
        if ATE_LOOP_COUNT & (1 << i):
            f = f * linefunc(r, q, p)
            r = pt_add(r, q)
    q1 = (q[0] ** Q, q[1] ** Q)
    nq2 = (q1[0] ** Q, -(q1[1] ** Q))
    f = f * linefunc(r, q1, p)
    r = pt_add(r, q1)
    f = f * linefunc(r, nq2, p)
    return f


def final_exponentiate(f):
    return f ** ((Q ** 12 - 1) // R)


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 for G1 points P_i (int pairs or None) and G2 points Q_i (Fq2 pairs or None): one final
    exponentiation for the whole product"""
    f = FQ12_ONE
    for p, q in pairs:
        if p is None or q is None:
            continue
        f = f * miller_loop(twist(q), cast_g1(p))
    return final_exponentiate(f) == FQ12_ONE
