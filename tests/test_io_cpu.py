"""Caller-side input formats (no GPU): .fvecs round trip and error behaviour (reference tests/common/mod.rs:104-124)."""
import numpy as np
import pytest

from halo2_vectordb_amd.io import read_fvecs, write_fvecs


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
