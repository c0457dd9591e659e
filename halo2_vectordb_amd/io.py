"""Input formats on the caller side of the path.

* `.fvecs` (SIFT / TEXMEX): per vector a little-endian u32 dimension followed by that many little-endian f32 — what the
  reference's `read_vectors_from_disk` parses (/root/reference/tests/common/mod.rs:104-124) for `test_siftsmall`
  (/root/reference/tests/demo_test.rs:59-88).  The dataset itself is not redistributed with the reference.
* `data/*.in` JSON: `{"a": [...], "b": [...]}` / `{"query": [...], "database": [[...], ...]}` style inputs of the
  examples (/root/reference/data, /root/reference/src/scaffold/cmd.rs) load with `json.load` as they are.
"""
import json
import struct

import numpy as np


def read_fvecs(path, count=None, dim=None):
    """Returns an (n, d) float64 array (the reference widens f32 to f64 before quantizing).  `count` limits the number
    of vectors read; `dim`, when given, is checked against every record's header like the reference does."""
    out = []
    with open(path, "rb") as f:
        while count is None or len(out) < count:
            head = f.read(4)
            if not head:
                break
            if len(head) != 4:
                raise ValueError("truncated fvecs record header")
            (d,) = struct.unpack("<I", head)
            if dim is not None and d != dim:
                raise ValueError(f"fvecs record has dimension {d}, expected {dim}")
            body = f.read(4 * d)
            if len(body) != 4 * d:
                raise ValueError("truncated fvecs record")
            out.append(np.frombuffer(body, dtype="<f4").astype(np.float64))
    if not out:
        return np.zeros((0, dim or 0), dtype=np.float64)
    if len({len(v) for v in out}) != 1:
        raise ValueError("fvecs records of different dimensions")
    return np.stack(out)


def write_fvecs(path, vectors):
    """Inverse of read_fvecs (used to build synthetic SIFT-shaped fixtures)."""
    v = np.asarray(vectors, dtype="<f4")
    with open(path, "wb") as f:
        for row in v:
            f.write(struct.pack("<I", len(row)))
            f.write(row.tobytes())


# ---------------------------------------------------------------- example inputs (data/*.in)
def load_input(path):
    """`data/{name}.in` of the reference's examples (serde_json of the example's `CircuitInput`): a JSON object whose
    values are numbers, vectors or lists of vectors.  Returned as numpy float64 arrays keyed like the file
    (e.g. {"vectors": (n, d)} for kmeans / merkle, {"query": (d,), "database": (n, d)} for query, {"a", "b"} for the
    distance examples)."""
    import json
    with open(path) as f:
        raw = json.load(f)
    return {k: np.asarray(v, dtype=np.float64) for k, v in raw.items()}


# ---------------------------------------------------------------- pinning file (configs/{name}.json)
# Keygen writes, Prove reads (src/scaffold/mod.rs:272, 285-287): the circuit shape parameters and the break points of
# the advice stream.  Layout recalled from snark-verifier-sdk's AggregationConfigPinning (the Pinning type of the
# scaffold, src/scaffold/mod.rs:436-438) — [UPSTREAM-RECALL, parity unpinned: the reference holds no sample file]:
#   {"params": {"degree": k, "num_advice": A, "num_lookup_advice": LA, "num_fixed": F, "lookup_bits": L},
#    "break_points": [[b_0, b_1, ...]]}          (one list per phase; this path has phase 0 only)
def write_pinning(path, k, break_points, num_lookup_advice, lookup_bits, num_fixed=1):
    import json
    bp = [int(b) for b in break_points]
    doc = {"params": {"degree": int(k), "num_advice": len(bp) + 1, "num_lookup_advice": int(num_lookup_advice), "num_fixed": int(num_fixed),
                      "lookup_bits": int(lookup_bits)},
           "break_points": [bp]}
    with open(path, "w") as f:
        json.dump(doc, f)


def read_pinning(path):
    """Returns (params dict, break points of phase 0 as uint64 array); checks the invariants the layout relies on."""
    import json
    with open(path) as f:
        doc = json.load(f)
    params, bps = doc["params"], doc["break_points"]
    if len(bps) < 1 or any(len(p) for p in bps[1:]):
        raise ValueError("only phase-0 break points are supported on this path")
    bp = np.asarray(bps[0], dtype=np.uint64)
    if params["num_advice"] != len(bp) + 1:
        raise ValueError("num_advice does not match the number of break points")
    rows = 1 << int(params["degree"])
    if len(bp) and int(bp.max()) >= rows:
        raise ValueError("break point beyond the column height")
    return params, bp


# ---------------------------------------------------------------- proof file (the reference writes data/{name}.snark, src/scaffold/mod.rs:292-297)
SNARK_MAGIC = b"VDBSNARK1\n"


def write_snark(path, proof, instances):
    """The Prove arm's output file in this build's own container (upstream's is a bincode dump of snark-verifier's Snark struct —
    protocol, instances, proof —, not reproduced: parity unpinned): magic, the number of public inputs, each as 32 little-endian
    bytes of the canonical value, then the proof bytes as the transcript wrote them."""
    with open(path, "wb") as f:
        f.write(SNARK_MAGIC)
        f.write(len(instances).to_bytes(4, "little"))
        for v in instances:
            f.write(int(v).to_bytes(32, "little"))
        f.write(len(proof).to_bytes(8, "little"))
        f.write(proof)


def read_snark(path):
    """-> (proof bytes, instances as integers); raises ValueError on a malformed file"""
    with open(path, "rb") as f:
        data = f.read()
    if not data.startswith(SNARK_MAGIC):
        raise ValueError("not a proof file of this build")
    pos = len(SNARK_MAGIC)
    if len(data) < pos + 4:
        raise ValueError("truncated proof file")
    n = int.from_bytes(data[pos: pos + 4], "little")
    pos += 4
    if len(data) < pos + 32 * n + 8:
        raise ValueError("truncated proof file")
    instances = [int.from_bytes(data[pos + 32 * i: pos + 32 * i + 32], "little") for i in range(n)]
    pos += 32 * n
    m = int.from_bytes(data[pos: pos + 8], "little")
    pos += 8
    if len(data) != pos + m:
        raise ValueError("proof length does not match the file")
    return data[pos:], instances


# ---------------------------------------------------------------- verifying key file (the reference writes data/{name}.vk, src/scaffold/mod.rs:276-281)
VK_FIXED = ("sel", "sigma", "cst", "table")


def write_verifying_key(path, meta, fixed):
    """meta: JSON-serialisable description of the circuit's shape (rounds.ProverRounds.save_verifying_key); fixed[name]: (n, 8) uint64
    commitments.  An .npz without pickled objects (numpy appends the suffix when `path` lacks it; pass a name ending in .npz)."""
    np.savez(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
             **{"fixed_" + name: np.ascontiguousarray(fixed[name], dtype=np.uint64).reshape(-1, 8) for name in VK_FIXED})


def read_verifying_key(path):
    """-> (meta dict with integers restored, {name: commitments}); raises ValueError on a malformed file"""
    try:
        with np.load(path, allow_pickle=False) as doc:
            meta = json.loads(bytes(doc["meta"]).decode())
            fixed = {name: np.ascontiguousarray(doc["fixed_" + name], dtype=np.uint64) for name in VK_FIXED}
    except (KeyError, OSError, json.JSONDecodeError, UnicodeDecodeError) as e:
        raise ValueError(f"not a verifying key of this build: {e}") from e
    for name, c in fixed.items():
        if c.ndim != 2 or c.shape[1] != 8:
            raise ValueError("verifying key: wrong shape for " + name)
    try:
        for key in ("delta", "tau", "vk_digest"):
            if key in meta:
                meta[key] = int(meta[key])
        meta["n_instances"] = int(meta["n_instances"])
        if "opened" in meta:
            meta["opened"] = {int(rot): list(names) for rot, names in meta["opened"].items()}
        shape = (meta["n_adv"], meta["n_cols"] + 2, 1, 1)        # sigma: advice, lookup, the constants' column, the instance column
    except (KeyError, TypeError, ValueError) as e:
        raise ValueError(f"verifying key: bad description: {e}") from e
    if tuple(len(fixed[name]) for name in VK_FIXED) != shape:
        raise ValueError("verifying key: the commitments do not match the described shape")
    return meta, fixed


# ---------------------------------------------------------------- halo2's own key files (SerdeFormat::RawBytes)
# What `pk.get_vk().write(&mut writer, SerdeFormat::RawBytes)` (/root/reference/src/scaffold/mod.rs:276-281), snark-verifier-sdk's
# gen_pk / read_pk (:273, :325-331) and `VerifyingKey::read(.., SerdeFormat::RawBytes)` (:334-343) put on disk.  The layout lives in
# halo2_proofs (plonk.rs `VerifyingKey::write`, `ProvingKey::write`; poly.rs `Polynomial::write`; helpers.rs `SerdeCurveAffine`,
# `pack`), a git dependency absent from /root/reference: it is restated here as published ([UPSTREAM-RECALL], parity unpinned —
# no key file ships with the reference to compare with):
#   verifying key   u32 BE k | u32 BE number of fixed commitments | the fixed commitments | the permutation's commitments (no count:
#                   the reader knows the columns from the circuit) | per selector 2^k bits, packed 8 to a byte, least significant first
#   point           x then y, each the four 64-bit limbs of the MONTGOMERY form, little-endian (RawBytes = the in-memory words; the
#                   identity is (0, 0)) — the layout of this build's (n, 8) uint64 commitment arrays as they are
#   polynomial      u32 BE number of values | the values, each four Montgomery limbs little-endian
#   polynomials     u32 BE count | that many polynomials
#   proving key     the verifying key | l_0, l_last, l_active_row (extended domain) | fixed columns: values (Lagrange), coefficient
#                   forms, extended cosets | permutation: sigma values (Lagrange), coefficient forms, extended cosets
# Fixed columns in the order halo2-base's RangeConfig::configure creates them ([UPSTREAM-RECALL]): the lookup table, the constants'
# column(s), then one fixed column per gate selector (selector compression cannot join two gate selectors: both are on in row 0).

def _be32(v):
    return int(v).to_bytes(4, "big")


def _take(f, n):
    b = f.read(n)
    if len(b) != n:
        raise ValueError("truncated key file")
    return b


def write_vk_raw(f, k, fixed_commitments, permutation_commitments, selectors):
    """f: binary file; commitments: (n, 8) uint64 Montgomery affine points; selectors: (n_selectors, 2^k) booleans (or an iterable
    of such blocks, for keys whose selectors are made a block of columns at a time)"""
    f.write(_be32(k))
    fixed = np.ascontiguousarray(fixed_commitments, dtype="<u8").reshape(-1, 8)
    f.write(_be32(len(fixed)))
    f.write(fixed.tobytes())
    f.write(np.ascontiguousarray(permutation_commitments, dtype="<u8").reshape(-1, 8).tobytes())
    for block in ([selectors] if isinstance(selectors, np.ndarray) else selectors):
        block = np.asarray(block, dtype=bool).reshape(-1, 1 << k)
        f.write(np.packbits(block, axis=1, bitorder="little").tobytes())


def read_vk_raw(f, n_permutation, n_selectors):
    """The reader needs what halo2's needs from the circuit's ConstraintSystem: how many columns the permutation argument covers
    and how many selectors there are.  -> dict(k, fixed_commitments, permutation_commitments, selectors)"""
    k = int.from_bytes(_take(f, 4), "big")
    if not 1 <= k <= 28:
        raise ValueError("verifying key: implausible k")
    n_fixed = int.from_bytes(_take(f, 4), "big")
    pts = lambda n: np.frombuffer(_take(f, 64 * n), dtype="<u8").reshape(n, 8).astype(np.uint64)
    fixed, perm = pts(n_fixed), pts(n_permutation)
    row_bytes = ((1 << k) + 7) // 8
    bits = np.frombuffer(_take(f, row_bytes * n_selectors), dtype=np.uint8).reshape(n_selectors, row_bytes)
    selectors = np.unpackbits(bits, axis=1, bitorder="little")[:, : 1 << k].astype(bool)
    return dict(k=k, fixed_commitments=fixed, permutation_commitments=perm, selectors=selectors)


def write_polys_raw(f, polys, count=None):
    """a Vec<Polynomial>: polys is (count, n, 4) uint64, or an iterable of (m, n, 4) blocks summing to `count` polynomials"""
    if isinstance(polys, np.ndarray):
        polys = np.ascontiguousarray(polys, dtype="<u8")
        count, polys = len(polys), [polys]
    f.write(_be32(count))
    done = 0
    for block in polys:
        block = np.ascontiguousarray(block, dtype="<u8")
        head = _be32(block.shape[1])
        for poly in block:
            f.write(head)
            f.write(poly.tobytes())
        done += len(block)
    if done != count:
        raise ValueError("polynomial count does not match what was announced")


def write_poly_raw(f, values):
    values = np.ascontiguousarray(values, dtype="<u8").reshape(-1, 4)
    f.write(_be32(len(values)))
    f.write(values.tobytes())


def read_poly_raw(f, expect=None):
    n = int.from_bytes(_take(f, 4), "big")
    if expect is not None and n != expect:
        raise ValueError(f"key file: a polynomial of {n} values where {expect} were expected")
    return np.frombuffer(_take(f, 32 * n), dtype="<u8").reshape(n, 4).astype(np.uint64)


def read_polys_raw(f, expect_count=None, expect=None, keep=True):
    """-> (count, n, 4) uint64; keep=False skips over the values (the extended cosets a loader re-derives) and returns the count"""
    count = int.from_bytes(_take(f, 4), "big")
    if expect_count is not None and count != expect_count:
        raise ValueError(f"key file: {count} polynomials where {expect_count} were expected")
    out = []
    for _ in range(count):
        if keep:
            out.append(read_poly_raw(f, expect))
        else:
            n = int.from_bytes(_take(f, 4), "big")
            if expect is not None and n != expect:
                raise ValueError(f"key file: a polynomial of {n} values where {expect} were expected")
            f.seek(32 * n, 1)
    if not keep:
        return count
    return np.stack(out) if out else np.zeros((0, expect or 0, 4), dtype=np.uint64)


def read_verifying_key_raw(path, n_instances=0, tau=None):
    """A data/{name}.vk in halo2's RawBytes layout -> (meta, fixed) as read_verifying_key returns them.  What halo2 takes from the
    circuit's ConstraintSystem this reader takes from the file's length and this build's circuit family: one constants' column and
    one range table, one selector per gate
    column, the permutation over [advice | lookup | constants | instance].  The lookup table's commitment sits first, as upstream."""
    import os
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        k = int.from_bytes(_take(f, 4), "big")
        n_fixed = int.from_bytes(_take(f, 4), "big")
        if not 1 <= k <= 28 or n_fixed < 3:
            raise ValueError("not a verifying key of this circuit family")
        n_adv = n_fixed - 2
        rest = size - 8 - 64 * n_fixed - n_adv * (((1 << k) + 7) // 8)
        if rest < 0 or rest % 64:
            raise ValueError("verifying key: the file's length does not fit its header")
        n_perm = rest // 64
        f.seek(0)
        doc = read_vk_raw(f, n_perm, n_adv)
    n_lk = n_perm - n_adv - 2
    if n_lk < 0:
        raise ValueError("verifying key: fewer permutation columns than gate columns")
    chunk_len = 2 if n_lk else 1
    from .rounds import N_BLIND, _fr_to_int
    from . import api
    meta = dict(rows=1 << k, k=k, n_adv=n_adv, n_lk=n_lk, n_cols=n_adv + n_lk, n_sets=-(-n_perm // chunk_len), chunk_len=chunk_len, n_blind=N_BLIND,
                delta=_fr_to_int(api.fr_delta()), n_instances=int(n_instances))
    if tau is not None:
        meta["tau"] = int(tau)
    names = {0: ["adv", "sel", "sigma", "cst", "table", "pa", "ps", "zp", "zl", "hf", "rand"], 1: ["advg", "zp", "zl"], 2: ["advg"], 3: ["advg"], -1: ["pa"], -N_BLIND: ["zp"]}
    lookup_only = {"pa", "ps", "zl"}
    meta["opened"] = {rot: [n for n in ns if n_lk or n not in lookup_only] for rot, ns in names.items()}
    meta["opened"] = {rot: ns for rot, ns in meta["opened"].items() if ns}
    fc = doc["fixed_commitments"]
    fixed = {"table": fc[0:1], "cst": fc[1:2], "sel": fc[2:], "sigma": doc["permutation_commitments"]}
    return meta, fixed, doc["selectors"]
