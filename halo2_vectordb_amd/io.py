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
