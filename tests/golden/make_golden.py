#!/usr/bin/env python3
"""Generates tests/golden/hotpath_golden.json: small committed input/output vectors for every C-ABI entry
point family.  Inputs are seeded; expected outputs come from the CPU oracle (oracle/liboracle.so) and are
cross-checked here against the independent Python implementation (oracle/pyref.py) where that is cheap.
The reference itself holds no expected outputs for this path and cannot be run, so these are
oracle-derived regression vectors, not reference-derived ones.  Run from the repo root."""
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


out = {}
# fixed-point staging: /root/reference/data/distances.in values
A, B = [0.123, 0.456, 1.789], [1.123, 0.456, 0.789]
qa, qb = O.quantize(A), O.quantize(B)
fp = PY.FixedPoint(48)
assert O.fr_to_ints(qa) == [fp.quantize(x) for x in A]
out["quantize"] = {"P": 48, "x": A + B, "q_mont_limbs": hx(np.concatenate([qa, qb]))}
# distances on that input, three metrics, L = 12 (README.md:66-79 runs distances with LOOKUP_BITS=12)
d = {}
for metric in ("euclidean", "cosine", "manhattan"):
    c = O.Ctx(store=True, keygen=True)
    r = c.distance(metric, qa, qb, L=12)
    ref = {"euclidean": fp.euclidean, "cosine": fp.cosine, "manhattan": fp.manhattan}[metric](O.fr_to_ints(qa), O.fr_to_ints(qb))
    assert O.fr_to_ints(r)[0] == ref
    d[metric] = {"result": hx(r)[0], "cells": len(c), "lookups": c.n_lookup, "stream_sha256": digest(c.advice()),
                 "lookup_sha256": digest(c.lookup()), "selector_sha256": digest(c.selectors())}
out["distance_distances_in_L12"] = d
# kmeans<2,2> on seeded vectors
rng = np.random.default_rng(4242)
vec = rng.random((6, 3))
qv = O.quantize(vec)
c = O.Ctx(store=True, keygen=True)
cent, ind = c.kmeans("euclidean", qv, 2, 2)
pc, pi = fp.kmeans([O.fr_to_ints(v) for v in qv], 2, 2, fp.euclidean)
assert [O.fr_to_ints(x) for x in cent] == pc
out["kmeans_6x3_K2_I2"] = {"seed": 4242, "vectors": vec.tolist(), "centroids": [hx(x) for x in cent], "cells": len(c), "lookups": c.n_lookup,
                           "stream_sha256": digest(c.advice()), "lookup_sha256": digest(c.lookup())}
# poseidon
P = PY.Poseidon()
msg = O.fr_from_ints([6, 100]).reshape(1, 2, 4)  # data/poseidon.in
h = O.poseidon_hash_many(msg)
assert O.fr_to_ints(h)[0] == P.hash([6, 100])
v = O.quantize(vec, 32)
root = O.poseidon_merkle_root(v)
assert O.fr_to_ints(root)[0] == P.merkle_root([O.fr_to_ints(x) for x in v])
c = O.Ctx(store=True)
c.merkle_commitment(v)
out["poseidon"] = {"hash_6_100": hx(h)[0], "merkle_root_6x3_P32": hx(root.reshape(1, 4))[0], "merkle_cells": len(c), "merkle_stream_sha256": digest(c.advice())}
# NTT
k = 6
col = O.random_fr(np.random.default_rng(77), 1 << k)
f = O.ntt(col, O.root_of_unity(k))
assert O.fr_to_ints(f) == PY.dft(O.fr_to_ints(col), O.fr_to_ints(O.root_of_unity(k))[0])
co, ex = O.lde_batch(col.reshape(1, 1 << k, 4), ext=2)
out["ntt_k6_seed77"] = {"forward_sha256": digest(f), "coeff_sha256": digest(co), "extended_sha256": digest(ex)}
# MSM
g, gl = O.srs_from_tau(5, 0xC0FFEE)
sc = O.random_fr(np.random.default_rng(78), 32)
pt = O.msm(sc, gl)
want = PY.msm(O.fr_to_ints(sc), [tuple(O.fq_to_ints(p.reshape(2, 4))) for p in gl])
assert tuple(O.fq_to_ints(pt.reshape(2, 4))) == want
out["msm_k5_tau_c0ffee_seed78"] = {"lagrange_commit": hx(pt.reshape(2, 4))}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "hotpath_golden.json"), "w"), indent=1)
print("wrote tests/golden/hotpath_golden.json")
