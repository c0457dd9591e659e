"""The reference's example workloads on its own input files (tests/golden/reference_data/*.in = /root/reference/data/*.in).

Each `oracle_*` function replays one example closure on the CPU restatement exactly as the example wires it
(examples/*.rs: which cells are assigned first, PRECISION_BITS, K, I, the distance closure) with the README's LOOKUP_BITS
and degree (README.md:66-79); each `f64_*` function is the plain floating-point computation the reference's tests compare
against (tests/distances/mod.rs:14-74, tests/vectordb/mod.rs:31-91, 202-218).  Shared by the CPU and the GPU test files.
"""
import json
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")

# README.md:66-79 (euclid: the commented-out line below them)
README = {
    "distances": dict(L=12, k=13), "euclid": dict(L=12, k=13), "merkle": dict(L=12, k=13), "query": dict(L=12, k=13),
    "kmeans": dict(L=15, k=16), "poseidon": dict(L=12, k=13),
}


def load(name):
    with open(os.path.join(DATA, name + ".in")) as f:
        return json.load(f)


# ------------------------------------------------------------------------------------------------ f64 side
def f64_euclidean(a, b):
    return float(np.sqrt(sum((x - y) ** 2 for x, y in zip(a, b))))


def f64_manhattan(a, b):
    return float(sum(abs(x - y) for x, y in zip(a, b)))


def f64_cosine(a, b):
    ab = sum(x * y for x, y in zip(a, b))
    aa = sum(x * x for x in a)
    bb = sum(y * y for y in b)
    return 1.0 - ab / (np.sqrt(aa) * np.sqrt(bb))


def f64_hamming(a, b):
    return 1.0 - sum(1.0 if x == y else 0.0 for x, y in zip(a, b)) / len(a)


F64 = dict(euclidean=f64_euclidean, manhattan=f64_manhattan, cosine=f64_cosine, hamming=f64_hamming)


def f64_kmeans(vectors, K, I, distance):
    """tests/vectordb/mod.rs:31-91 (first minimum wins; the distance closure is called as distance(v, c))"""
    vectors = [list(map(float, v)) for v in vectors]
    n = len(vectors[0])
    cent = [list(v) for v in vectors[:K]]
    ids = [0] * len(vectors)
    for _ in range(I):
        sizes = [0] * K
        for i, v in enumerate(vectors):
            d = [distance(v, c) for c in cent]
            ids[i] = d.index(min(d))
            sizes[ids[i]] += 1
        for k in range(K):
            mean = [0.0] * n
            for i, v in enumerate(vectors):
                if ids[i] == k:
                    for j in range(n):
                        mean[j] += v[j]
            cent[k] = [m / sizes[k] for m in mean]
    return np.array(cent), ids


def f64_nearest(query, vectors, distance):
    """tests/vectordb/mod.rs:202-218 (distance(v, query); first minimum)"""
    d = [distance(v, query) for v in vectors]
    i = d.index(min(d))
    return i, np.array(vectors[i], dtype=np.float64)


def rel_close(a, b, eps=1e-6):
    """assert_float_relative_eq! of the reference's tests: |a - b| <= eps * max(|a|, |b|)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all(np.abs(a - b) <= eps * np.maximum(np.abs(a), np.abs(b))))


# ------------------------------------------------------------------------------------------------ oracle side
def oracle_distances(O, name="distances"):
    """examples/distances.rs:27-59 (euclid: examples/euclid.rs:27-44, ten euclidean distances of the same pair)"""
    d, cfg = load(name), README[name]
    qa, qb = O.quantize(d["a"]), O.quantize(d["b"])
    c = O.Ctx(store=True, keygen=True, plan_k=cfg["k"])
    c.assign_witnesses(qa)
    c.assign_witnesses(qb)
    out = {"ctx": c, "qa": qa, "qb": qb, "offsets": {}, "results": {}}
    metrics = ["euclidean", "manhattan", "cosine", "hamming"] if name == "distances" else ["euclidean"] * 10
    for i, m in enumerate(metrics):
        out["offsets"][(m, i)] = (len(c), c.n_lookup)
        out["results"][(m, i)] = c.distance(m, qa, qb, L=cfg["L"])
    return out


def oracle_kmeans(O, name="kmeans", K=4, I=10, metric="cosine"):
    """examples/kmeans.rs:32-49: PRECISION_BITS 48, K 4, I 10, cosine distance (the FIXME at :48), vectors assigned first"""
    d, cfg = load(name), README["kmeans"]
    qv = O.quantize(np.array(d["vectors"], dtype=np.float64))
    c = O.Ctx(store=True, keygen=True, plan_k=cfg["k"])
    c.assign_witnesses(qv)
    off = (len(c), c.n_lookup)
    cent, ind = c.kmeans(metric, qv, K, I, P=48, L=cfg["L"])
    return dict(ctx=c, qv=qv, offset=off, centroids=cent, indicators=ind)


def oracle_merkle(O):
    """examples/merkle.rs:34-49: PRECISION_BITS 32, PoseidonChip::new before the vectors are assigned"""
    d, cfg = load("merkle"), README["merkle"]
    qv = O.quantize(np.array(d["vectors"], dtype=np.float64), 32)
    c = O.Ctx(store=True, keygen=True, plan_k=cfg["k"])
    c.poseidon_chip_new()
    c.assign_witnesses(qv)
    off = len(c)
    root = c.merkle_commitment(qv)
    return dict(ctx=c, qv=qv, offset=off, root=root)


def oracle_query(O):
    """examples/query.rs:40-72: PRECISION_BITS 48, PoseidonChip::new, query then database assigned, nearest_vector with the
    cosine distance, then the Merkle commitment of the database"""
    d, cfg = load("query"), README["query"]
    qq = O.quantize(np.array(d["query"], dtype=np.float64))
    qdb = O.quantize(np.array(d["database"], dtype=np.float64))
    c = O.Ctx(store=True, keygen=True, plan_k=cfg["k"])
    c.poseidon_chip_new()
    c.assign_witnesses(qq)
    c.assign_witnesses(qdb)
    off_nv = (len(c), c.n_lookup)
    ind, res = c.nearest_vector("cosine", qq, qdb, P=48, L=cfg["L"])
    off_mk = len(c)
    root = c.merkle_commitment(qdb)
    return dict(ctx=c, qq=qq, qdb=qdb, off_nv=off_nv, off_mk=off_mk, indicator=ind, result=res, root=root)


def oracle_poseidon(O):
    """examples/poseidon.rs:27-36: two loaded witnesses, PoseidonChip::new, update([x, y]), squeeze"""
    d = load("poseidon")
    xy = O.fr_from_ints([int(s) for s in d["inputs"]])
    c = O.Ctx(store=True, keygen=True, plan_k=README["poseidon"]["k"])
    c.assign_witnesses(xy)
    c.poseidon_chip_new()
    off = len(c)
    h = c.merkle_commitment(xy.reshape(1, 2, 4))   # one leaf and no tree level: exactly update([x, y]); squeeze
    return dict(ctx=c, xy=xy, offset=off, hash=h)
