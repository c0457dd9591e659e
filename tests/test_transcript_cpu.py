"""Fiat–Shamir transcript (host code of the product library, no GPU involved) against the independent
Python restatement of the Poseidon sponge in oracle/pyref.py, whose width-5 permutation is pinned to the Poseidon authors' published
vector poseidonperm_x5_254_5 (tests/test_oracle_poseidon.py).  Parity unpinned beyond that: that upstream uses this width and these
round numbers, the sponge's framing and the encodings are [UPSTREAM-RECALL] (the reference holds no transcript or proof bytes)."""
import numpy as np
import pytest

Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    return a


def _sponge(P, R, t, msgs_and_squeezes):
    """the documented rule on python ints: buffered absorb, RATE per permutation, +1 after a short chunk, an extra
    permutation when the buffer length is a multiple of RATE; state persists over squeezes"""
    rate = t - 1
    st = [1 << 64] + [0] * rate
    outs = []
    for msg in msgs_and_squeezes:
        chunks = [msg[i:i + rate] for i in range(0, len(msg), rate)]
        if len(msg) % rate == 0:
            chunks.append([])
        for ch in chunks:
            for i, v in enumerate(ch):
                st[1 + i] = (st[1 + i] + v) % R
            if len(ch) < rate:
                st[1 + len(ch)] = (st[1 + len(ch)] + 1) % R
            st = P.permute(st)
        outs.append(st[1])
    return outs


@pytest.mark.parametrize("t,r_f,r_p", [(5, 8, 60), (3, 8, 57), (4, 8, 56)])
def test_sponge_matches_python_restatement(api, O, PY, t, r_f, r_p):
    rng = np.random.default_rng(t)
    P = PY.Poseidon(t, r_f, r_p)
    batches = [[int(v) for v in rng.integers(1, 1 << 62, size=m)] for m in (0, 1, t - 1, t, 2 * (t - 1), 7, 0)]
    tr = api.Transcript(t, r_f, r_p)
    got = []
    for msg in batches:
        for v in msg:
            tr.common_scalar(O.fr_from_ints([v])[0])
        got.append(O.fr_to_ints(tr.squeeze().reshape(1, 4))[0])
    tr.free()
    assert got == _sponge(P, O.R_MOD, t, batches)


def test_width_three_sponge_is_the_chip_hash(api, O, PY):
    """one squeeze of a fresh width-3 transcript == PoseidonChip's hash of the same message (oracle C restatement, which
    runs the optimised schedule): the plain and the optimised permutations agree"""
    rng = np.random.default_rng(8)
    for m in (1, 2, 5, 8):
        msg = O.random_fr(rng, m)
        tr = api.Transcript(3, 8, 57)
        for v in msg:
            tr.common_scalar(v)
        got = tr.squeeze()
        tr.free()
        assert np.array_equal(got, O.poseidon_hash_many(msg.reshape(1, m, 4))[0])


def test_points_and_proof_bytes(api, O, PY):
    pts = O.g1_mul_generator([5, 123456789, 0])
    tr = api.Transcript()
    for p in pts:
        tr.write_point(p)
    s = O.fr_from_ints([0xDEADBEEF12345])[0]
    tr.write_scalar(s)
    ch = O.fr_to_ints(tr.squeeze().reshape(1, 4))[0]
    proof = tr.proof()
    tr.free()
    assert len(proof) == 4 * 32
    absorbed = []
    for i, p in enumerate(pts):
        x, y = O.fq_to_ints(p.reshape(2, 4))
        absorbed += [x % O.R_MOD, y % O.R_MOD]
        enc = int.from_bytes(proof[32 * i: 32 * i + 32], "little")
        assert enc & ((1 << 254) - 1) == x and (enc >> 254) == ((y & 1) if (x or y) else 0)
        if x or y:
            assert (y * y - x * x * x - 3) % Q_MOD == 0
    assert int.from_bytes(proof[96:], "little") == 0xDEADBEEF12345
    absorbed.append(0xDEADBEEF12345)
    assert ch == _sponge(PY.Poseidon(5, 8, 60), O.R_MOD, 5, [absorbed])[0]
    # bit 6 of the last byte is bit 254 of the little-endian integer
    assert all((proof[32 * i + 31] >> 7) == 0 for i in range(4))


def test_compressed_point_sign_bit_is_selectable():
    """the y-parity flag of a written point: bit 6 by default, bit 7 on request (halo2curves 0.3.x as recalled); the challenge does
    not depend on the byte encoding"""
    import numpy as np
    from halo2_vectordb_amd import api
    g = np.zeros(8, dtype=np.uint64)                  # the generator (1, 2) in Montgomery form has an even y; use (1, -2): odd y
    q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    mont = lambda v: [(v * (1 << 256) % q >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    g[:4], g[4:] = mont(1), mont(q - 2)
    out = {}
    for bit in (None, 6, 7):
        tr = api.Transcript()
        if bit is not None:
            tr.set_sign_bit(bit)
        tr.write_point(g)
        out[bit] = (tr.proof(), tr.squeeze().tobytes())
        tr.free()
    assert out[None] == out[6] and out[6][1] == out[7][1]
    assert out[6][0][31] == 0x40 and out[7][0][31] == 0x80 and out[6][0][:31] == out[7][0][:31]
    with pytest.raises(api.VdbError):
        tr = api.Transcript()
        tr.set_sign_bit(5)


def test_the_host_builds_of_the_sponge_agree():
    """The sponge's permutation exists in three host builds (portable, BMI2 + ADX, AVX-512 IFMA); the library picks by CPU feature and a
    one-time measurement, VDB_HOST_GENERIC=1 forces the portable one, 2 allows at most the mulx build, 3 takes the IFMA build whenever
    the CPU has it: the same absorbed values give the same challenges and the same Horner value in all, at widths that fit the vector
    build (3, 5, 8) and one that does not (9: the scalar builds serve).  On a CPU without a feature the runs coincide trivially."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import numpy as np, ctypes\n"
            "from halo2_vectordb_amd import api, _lib\n"
            "rng = np.random.default_rng(5)\n"
            "v = rng.integers(0, 1 << 62, size=(1001, 4), dtype=np.uint64)\n"
            "out = []\n"
            "for t, rp in ((5, 60), (3, 57), (8, 22), (9, 22)):\n"
            "    tr = api.Transcript(t, 8, rp)\n"
            "    tr.write_scalars(v)\n"
            "    a = tr.squeeze(); b = tr.squeeze()\n"
            "    out += list(a) + list(b)\n"
            "acc = np.zeros(4, dtype=np.uint64)\n"
            "_lib.check(_lib.load().vdb_fr_horner(api._p(v), ctypes.c_size_t(len(v)), api._p(a), api._p(acc)))\n"
            "print(' '.join(str(int(x)) for x in out + list(acc)))\n")
    outs = []
    for force in ("0", "1", "2", "3"):
        env = dict(os.environ, VDB_HOST_GENERIC=force, PYTHONPATH=root)
        outs.append(subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300, cwd=root))
        assert outs[-1].returncode == 0, outs[-1].stderr
    assert outs[0].stdout == outs[1].stdout == outs[2].stdout == outs[3].stdout and len(outs[0].stdout.split()) == 4 * 8 + 4


def test_flush_absorbs_early_without_changing_the_sponge():
    """vdb_transcript_flush takes the complete chunks now: challenges equal those of a transcript that leaves everything to the squeeze,
    whatever the lengths written between flushes (multiples of the rate, remainders, nothing)"""
    from halo2_vectordb_amd import api
    rng = np.random.default_rng(9)
    v = rng.integers(0, 1 << 62, size=(41, 4), dtype=np.uint64)
    a, b = api.Transcript(), api.Transcript()
    got_a, got_b = [], []
    pos = 0
    for m in (0, 3, 4, 1, 8, 5, 0, 9, 11):
        a.write_scalars(v[pos: pos + m])
        b.write_scalars(v[pos: pos + m])
        b.flush()
        b.flush()
        pos += m
        if m in (4, 5, 0, 11):
            got_a.append(a.squeeze().tolist())
            got_b.append(b.squeeze().tolist())
    assert got_a == got_b and a.proof() == b.proof()
    a.free()
    b.free()
