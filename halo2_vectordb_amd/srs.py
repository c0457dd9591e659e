"""The reference's "unsafe" universal setup, reproduced deterministically.

`gen_srs(k)` (/root/reference/src/scaffold/mod.rs:260, halo2-base `utils::fs::gen_srs`) creates
`ParamsKZG::<Bn256>::setup(k, ChaCha20Rng::from_seed(Default::default()))` when no params file exists; `setup` draws
ONE scalar, `s = Fr::random(rng)`, and derives g[i] = [s^i] G1, g_lagrange, [s] G2 from it.  With the all-zero seed
that scalar is a constant:

    Fr::random(rng) = Fr::from_u512([rng.next_u64(); 8])   = (first 64 keystream bytes, little-endian) mod r

[UPSTREAM-RECALL for the call chain — parity unpinned: the reference ships no params file to compare with; the RNG
itself is pinned by the RFC 7539 zero-key keystream vectors in tests/test_io_cpu.py.]  The powers are then computed on
the GPU by vdb_srs_setup_unsafe.
"""
import struct

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _rotl(v, c):
    return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))


def _quarter(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = _rotl(s[b] ^ s[c], 7)


def chacha20_block(key32, counter, stream=0):
    """One 64-byte ChaCha20 block in rand_chacha's layout: 64-bit block counter (words 12-13), 64-bit stream id
    (words 14-15); identical to RFC 7539 for stream 0 and counters below 2^32."""
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(struct.unpack("<8I", key32))
    st += [counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, stream & 0xFFFFFFFF, (stream >> 32) & 0xFFFFFFFF]
    w = st[:]
    for _ in range(10):
        _quarter(w, 0, 4, 8, 12); _quarter(w, 1, 5, 9, 13); _quarter(w, 2, 6, 10, 14); _quarter(w, 3, 7, 11, 15)
        _quarter(w, 0, 5, 10, 15); _quarter(w, 1, 6, 11, 12); _quarter(w, 2, 7, 8, 13); _quarter(w, 3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & 0xFFFFFFFF for a, b in zip(w, st)])


class ChaCha20Rng:
    """rand_chacha::ChaCha20Rng::from_seed(seed): sequential keystream, next_u64 = two little-endian u32 words."""

    def __init__(self, seed=bytes(32)):
        self.key, self.counter, self.buf = bytes(seed), 0, b""

    def fill_bytes(self, n):
        while len(self.buf) < n:
            self.buf += chacha20_block(self.key, self.counter)
            self.counter += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out

    def next_u64(self):
        return int.from_bytes(self.fill_bytes(8), "little")


def fr_random(rng):
    """halo2curves `Fr::random`: from_u512 of eight next_u64 limbs (little-endian), reduced mod r."""
    limbs = [rng.next_u64() for _ in range(8)]
    return sum(l << (64 * i) for i, l in enumerate(limbs)) % R_MOD


def gen_srs_tau(seed=bytes(32)):
    """The toxic-waste scalar of `gen_srs` (canonical integer)."""
    return fr_random(ChaCha20Rng(seed))


def tau_mont_limbs(tau):
    """canonical integer -> Montgomery form as 4 x u64 (the layout vdb_srs_setup_unsafe takes)"""
    v = tau * (1 << 256) % R_MOD
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
