#!/usr/bin/env python3
"""Generates tests/golden/rounds_golden.json: small committed vectors for the prover-round entry points.
Inputs are seeded; expected outputs come from the CPU restatements (oracle/oracle.py on the C oracle's field arithmetic, and
oracle/pyref.py for the transcript sponge).  The reference holds no expected outputs for these rounds and cannot
be run: these are regression vectors derived from the restatements, not from the reference.  Run from the repo root."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import pyref as PY  # noqa: E402


def hx(a):
    return ["%064x" % v for v in O.limbs_to_ints(a)]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def inputs(seed):
    """the seeded inputs every consumer of the file rebuilds"""
    rng = np.random.default_rng(seed)
    k, n_cols, usable, bits = 5, 4, 26, 3
    n = 1 << k
    cols = O.fr_from_ints([int(v) for v in rng.integers(0, 5, size=n_cols * n)]).reshape(n_cols, n, 4)
    mapping = np.array([[(c << 32) | r for r in range(n)] for c in range(n_cols)], dtype=np.uint64)
    ints = O.fr_to_ints(cols.reshape(-1, 4))
    by_val = {}
    for c in range(n_cols):
        for r in range(usable):
            by_val.setdefault(ints[c * n + r], []).append((c, r))
    for cells in by_val.values():
        for (c, r), (c2, r2) in zip(cells, cells[1:] + cells[:1]):
            mapping[c, r] = (c2 << 32) | r2
    beta, gamma, x = O.random_fr(rng, 1)[0], O.random_fr(rng, 1)[0], O.random_fr(rng, 1)[0]
    table = O.fr_from_ints([i if i < (1 << bits) else 0 for i in range(n)])
    lk = O.fr_from_ints([int(v) for v in rng.integers(0, 1 << bits, size=2 * n)]).reshape(2, n, 4)
    polys = O.random_fr(rng, 3 * n).reshape(3, n, 4)
    return dict(k=k, n=n, usable=usable, bits=bits, cols=cols, mapping=mapping, beta=beta, gamma=gamma, x=x, table=table, lk=lk, polys=polys)


def expected(I):
    n, usable = I["n"], I["usable"]
    sigma = O.permutation_sigma(I["mapping"], I["k"])
    zp = O.permutation_product(I["cols"], sigma, usable, 3, I["beta"], I["gamma"])
    tab = O.fr_to_ints(I["table"][:usable])
    pa, ps = [], []
    for c in range(2):
        a, s = O.lookup_permute(O.fr_to_ints(I["lk"][c, :usable]), tab)
        pa.append(O.fr_from_ints(a + [0] * (n - usable)))
        ps.append(O.fr_from_ints(s + [0] * (n - usable)))
    pa, ps = np.stack(pa), np.stack(ps)
    zl = O.lookup_product(I["lk"], I["table"], pa, ps, usable, I["beta"], I["gamma"])
    ev = O.eval_polys(I["polys"], I["x"])
    R = O.R_MOD
    xi = O.fr_to_ints(I["x"].reshape(1, 4))[0]
    quot = []
    for c in range(3):
        a = O.fr_to_ints(I["polys"][c])
        q, s = [0] * n, 0
        for i in range(n - 1, 0, -1):
            s = (a[i] + xi * s) % R
            q[i - 1] = s
        quot.append(O.fr_from_ints(q))
    return dict(sigma=sigma, zp=zp, pa=pa, ps=ps, zl=zl, evals=ev, kate=np.stack(quot))


if __name__ == "__main__":
    seed = 20261004
    I = inputs(seed)
    E = expected(I)
    one = O.fr_from_ints([1])[0]
    assert np.array_equal(E["zp"][-1, I["usable"]], one) and all(np.array_equal(z[I["usable"]], one) for z in E["zl"])
    out = {"seed": seed, "sha256": {name: digest(v) for name, v in E.items()}, "evals": hx(E["evals"]), "zp_last_row": hx(E["zp"][:, I["usable"]])}
    # transcript: width 5, 8 + 60 rounds; absorb 1..9, squeeze, absorb 10, 11, squeeze, squeeze
    P = PY.Poseidon(5, 8, 60)
    st = [1 << 64, 0, 0, 0, 0]
    sq = []
    for msg in ([1, 2, 3, 4, 5, 6, 7, 8, 9], [10, 11], []):
        chunks = [msg[i:i + 4] for i in range(0, len(msg), 4)] + ([[]] if len(msg) % 4 == 0 else [])
        for ch in chunks:
            for i, v in enumerate(ch):
                st[1 + i] = (st[1 + i] + v) % O.R_MOD
            if len(ch) < 4:
                st[1 + len(ch)] = (st[1 + len(ch)] + 1) % O.R_MOD
            st = P.permute(st)
        sq.append("%064x" % st[1])
    out["transcript_t5_squeezes"] = sq
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "rounds_golden.json"), "w"), indent=1)
    print("wrote tests/golden/rounds_golden.json")
