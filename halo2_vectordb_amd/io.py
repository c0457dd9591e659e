"""Input formats on the caller side of the path.

* `.fvecs` (SIFT / TEXMEX): per vector a little-endian u32 dimension followed by that many little-endian f32 — what the
  reference's `read_vectors_from_disk` parses (/root/reference/tests/common/mod.rs:104-124) for `test_siftsmall`
  (/root/reference/tests/demo_test.rs:59-88).  The dataset itself is not redistributed with the reference.
* `data/*.in` JSON: `{"a": [...], "b": [...]}` / `{"query": [...], "database": [[...], ...]}` style inputs of the
  examples (/root/reference/data, /root/reference/src/scaffold/cmd.rs) load with `json.load` as they are.
"""
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
