"""Caller-side input formats (no GPU): .fvecs round trip and error behaviour (reference tests/common/mod.rs:104-124)."""
import numpy as np
import pytest

from halo2_vectordb_amd.io import load_input, read_fvecs, read_pinning, write_fvecs, write_pinning


def test_fvecs_roundtrip(tmp_path):
    rng = np.random.default_rng(1)
    v = rng.integers(0, 219, size=(17, 128)).astype(np.float64)
    p = tmp_path / "base.fvecs"
    write_fvecs(p, v)
    assert p.stat().st_size == 17 * (4 + 4 * 128)
    assert np.array_equal(read_fvecs(p), v)
    assert np.array_equal(read_fvecs(p, count=10, dim=128), v[:10])
    assert read_fvecs(p, count=0).shape[0] == 0


def test_fvecs_errors(tmp_path):
    p = tmp_path / "bad.fvecs"
    write_fvecs(p, np.ones((2, 4)))
    with pytest.raises(ValueError):
        read_fvecs(p, dim=8)
    data = p.read_bytes()
    p.write_bytes(data[:-3])
    with pytest.raises(ValueError):
        read_fvecs(p)
    e = tmp_path / "empty.fvecs"
    e.write_bytes(b"")
    assert read_fvecs(e).shape[0] == 0


def test_pinning_roundtrip_and_checks(tmp_path):
    p = tmp_path / "kmeans.json"
    bp = [65526, 65527, 65520]
    write_pinning(p, 16, bp, num_lookup_advice=2, lookup_bits=15)
    params, got = read_pinning(p)
    assert params == {"degree": 16, "num_advice": 4, "num_lookup_advice": 2, "num_fixed": 1, "lookup_bits": 15}
    assert got.dtype == np.uint64 and got.tolist() == bp
    import json
    doc = json.loads(p.read_text())
    doc["params"]["num_advice"] = 9
    p.write_text(json.dumps(doc))
    with pytest.raises(ValueError):
        read_pinning(p)
    doc["params"]["num_advice"] = 4
    doc["break_points"][0][0] = 1 << 16
    p.write_text(json.dumps(doc))
    with pytest.raises(ValueError):
        read_pinning(p)


def test_load_example_inputs(tmp_path):
    """the shapes of the reference's data/*.in files (data/kmeans.in, data/query.in, data/distances.in)"""
    (tmp_path / "kmeans.in").write_text('{"vectors": [[1.123, 0.456, 0.789], [1.111, 0.111, 0.111], [0.111, 0.444, 1.777]]}')
    (tmp_path / "query.in").write_text('{"query": [0.123, 0.456, 1.789], "database": [[1.123, 0.456, 0.789], [1.111, 0.111, 0.111]]}')
    (tmp_path / "distances.in").write_text('{"a": [0.123, 0.456, 1.789], "b": [1.123, 0.456, 0.789]}')
    assert load_input(tmp_path / "kmeans.in")["vectors"].shape == (3, 3)
    q = load_input(tmp_path / "query.in")
    assert q["query"].shape == (3,) and q["database"].shape == (2, 3)
    d = load_input(tmp_path / "distances.in")
    assert d["a"].tolist() == [0.123, 0.456, 1.789] and d["b"][0] == 1.123
