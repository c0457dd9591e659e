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


def test_chacha20_rng_and_srs_scalar():
    """rand_chacha::ChaCha20Rng::from_seed([0; 32]) against the RFC 7539 appendix A.1 zero-key keystream (blocks 0 and 1),
    and the scalar gen_srs derives from it"""
    from halo2_vectordb_amd.srs import ChaCha20Rng, R_MOD, chacha20_block, fr_random, gen_srs_tau
    b0 = bytes.fromhex("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                       "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    b1 = bytes.fromhex("9f07e7be5551387a98ba977c732d080dcb0f29a048e3656912c6533e32ee7aed"
                       "29b721769ce64e43d57133b074d839d531ed1f28510afb45ace10a1f4b794d6f")
    assert chacha20_block(bytes(32), 0) == b0 and chacha20_block(bytes(32), 1) == b1
    rng = ChaCha20Rng()
    assert rng.next_u64() == int.from_bytes(b0[:8], "little")      # rand_chacha: first words 0xade0b876, 0x903df1a0
    assert rng.next_u64() == int.from_bytes(b0[8:16], "little")
    tau = gen_srs_tau()
    assert tau == int.from_bytes(b0, "little") % R_MOD and 0 < tau < R_MOD
    rng2 = ChaCha20Rng()
    assert fr_random(rng2) == tau and fr_random(rng2) == int.from_bytes(b1, "little") % R_MOD
    # RFC 7539 2.3.2 block-function test vector (key 00..1f, counter 1, nonce 000000090000004a00000000)
    key = bytes(range(32))
    nonce_as_stream = int.from_bytes(bytes.fromhex("0000004a00000000"), "little")
    blk = chacha20_block(key, 1 | (0x09000000 << 32), nonce_as_stream)
    assert blk[:16].hex() == "10f1e7e4d13b5915500fdd1fa32071c4"


def test_snark_file_round_trip(tmp_path):
    from halo2_vectordb_amd.io import read_snark, write_snark
    proof, inst = bytes(range(256)) * 3, [5, (1 << 253) + 17]
    write_snark(tmp_path / "p.snark", proof, inst)
    assert read_snark(tmp_path / "p.snark") == (proof, inst)
    write_snark(tmp_path / "e.snark", b"", [])
    assert read_snark(tmp_path / "e.snark") == (b"", [])
    raw = (tmp_path / "p.snark").read_bytes()
    for bad in (raw[:-1], raw + b"x", b"junk" + raw, raw[:12]):
        (tmp_path / "bad.snark").write_bytes(bad)
        with pytest.raises(ValueError):
            read_snark(tmp_path / "bad.snark")


def test_verifying_key_file_round_trip(tmp_path):
    """io.write_verifying_key / read_verifying_key (the Keygen arm's .vk, src/scaffold/mod.rs:276-281, in this build's container):
    integers restored, shapes checked against the description, malformed files refused with ValueError"""
    import numpy as np
    import pytest
    from halo2_vectordb_amd.io import VK_FIXED, read_verifying_key, write_verifying_key
    rng = np.random.default_rng(3)
    n_adv, n_cols = 5, 7
    counts = dict(sel=n_adv, sigma=n_cols + 2, cst=1, table=1)
    fixed = {name: rng.integers(0, 1 << 63, size=(counts[name], 8), dtype=np.uint64) for name in VK_FIXED}
    meta = dict(rows=64, k=6, n_adv=n_adv, n_lk=2, n_cols=n_cols, n_sets=3, chunk_len=3, n_blind=5, delta=str(7 ** 40), n_instances=2,
                tau=str(2 ** 200 + 5), vk_digest=str(3 ** 150), opened={"0": ["adv", "sel"], "-1": ["pa"]})
    path = str(tmp_path / "c.vk.npz")
    write_verifying_key(path, meta, fixed)
    m, f = read_verifying_key(path)
    assert m["delta"] == 7 ** 40 and m["tau"] == 2 ** 200 + 5 and m["vk_digest"] == 3 ** 150 and m["n_instances"] == 2
    assert m["opened"] == {0: ["adv", "sel"], -1: ["pa"]} and all(np.array_equal(f[name], fixed[name]) for name in VK_FIXED)
    # a key whose commitments do not fit its description, a file that is something else
    write_verifying_key(path, dict(meta, n_adv=n_adv + 1), fixed)
    with pytest.raises(ValueError):
        read_verifying_key(path)
    np.savez(path, meta=np.frombuffer(b"{}", dtype=np.uint8))
    with pytest.raises(ValueError):
        read_verifying_key(path)
    with pytest.raises(ValueError):
        read_verifying_key(str(tmp_path / "absent.npz"))


def test_halo2_rawbytes_key_layout_round_trips():
    """io.write_vk_raw / read_vk_raw / write_polys_raw / read_polys_raw: halo2's SerdeFormat::RawBytes key layout as restated in io.py
    ([UPSTREAM-RECALL], parity unpinned: the reference ships no key file) — big-endian u32 counts, Montgomery limbs little-endian,
    selector bits packed least significant first"""
    import io as _io
    from halo2_vectordb_amd import io as vio
    rng = np.random.default_rng(5)
    k = 4
    fixed = rng.integers(0, 1 << 63, (3, 8), dtype=np.uint64)
    perm = rng.integers(0, 1 << 63, (5, 8), dtype=np.uint64)
    sel = np.zeros((2, 16), dtype=bool)
    sel[0, [0, 3, 9]] = True
    sel[1, 15] = True
    f = _io.BytesIO()
    vio.write_vk_raw(f, k, fixed, perm, sel)
    raw = f.getvalue()
    assert raw[:8] == bytes([0, 0, 0, 4, 0, 0, 0, 3]) and len(raw) == 8 + 64 * 8 + 2 * 2
    assert raw[8:16] == int(fixed[0, 0]).to_bytes(8, "little")
    assert raw[-4:] == bytes([0b00001001, 0b00000010, 0, 0b10000000])
    f.seek(0)
    doc = vio.read_vk_raw(f, 5, 2)
    assert doc["k"] == k and np.array_equal(doc["fixed_commitments"], fixed) and np.array_equal(doc["permutation_commitments"], perm)
    assert np.array_equal(doc["selectors"], sel)
    # the selectors in blocks give the same bytes
    g = _io.BytesIO()
    vio.write_vk_raw(g, k, fixed, perm, iter([sel[:1], sel[1:]]))
    assert g.getvalue() == raw
    with pytest.raises(ValueError):
        vio.read_vk_raw(_io.BytesIO(raw[:-1]), 5, 2)
    polys = rng.integers(0, 1 << 63, (3, 16, 4), dtype=np.uint64)
    f = _io.BytesIO()
    vio.write_polys_raw(f, iter([polys[:2], polys[2:]]), 3)
    raw = f.getvalue()
    assert raw[:8] == bytes([0, 0, 0, 3, 0, 0, 0, 16]) and len(raw) == 4 + 3 * (4 + 16 * 32)
    f.seek(0)
    assert np.array_equal(vio.read_polys_raw(f, 3, 16), polys)
    f.seek(0)
    assert vio.read_polys_raw(f, 3, 16, keep=False) == 3 and not f.read(1)
    with pytest.raises(ValueError):
        vio.read_polys_raw(_io.BytesIO(raw), 4, 16)
    with pytest.raises(ValueError):
        vio.write_polys_raw(_io.BytesIO(), iter([polys[:2]]), 3)
