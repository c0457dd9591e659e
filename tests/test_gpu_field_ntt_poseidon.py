"""GPU parity (through the C ABI) of the field helpers, the NTT family and Poseidon against the CPU
oracle on the same seeded inputs.  Bit-exact: integer arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init()
    return a


def test_native_library_is_loaded(api):
    from halo2_vectordb_amd import _lib
    assert _lib.load().vdb_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libvdb_hip.so" in maps


def test_field_ops(api, O):
    rng = np.random.default_rng(21)
    a, b = O.random_fr(rng, 5000), O.random_fr(rng, 5000)
    edge = O.fr_from_ints([0, 1, R - 1, R - 2, 1 << 48, R - (1 << 97), 1 << 253, 2])
    a[:8], b[:8] = edge, edge[::-1]
    assert np.array_equal(api.fr_mul(a, b), O.fr_mul(a, b))
    assert np.array_equal(api.fr_add(a, b), O.fr_add(a, b))
    assert np.array_equal(api.fr_sub(a, b), O.fr_sub(a, b))
    assert np.array_equal(api.fr_to_canonical(a), O.fr_to_canonical(a))
    can = O.fr_to_canonical(a)
    assert np.array_equal(api.fr_from_canonical(can), a)
    assert np.array_equal(api.fr_mul(a[:0], b[:0]), a[:0])  # empty input


def test_batch_invert(api, O):
    rng = np.random.default_rng(22)
    a = O.random_fr(rng, 1000)
    a[[0, 5, 31, 32, 999]] = 0
    got = api.fr_batch_invert(a)
    assert np.array_equal(got, O.fr_inv(a))  # oracle maps 0 -> 0 as well


@pytest.mark.parametrize("k", [0, 1, 2, 5, 8, 10, 11, 12, 14, 16, 17])
def test_ntt_forward(api, O, k):
    rng = np.random.default_rng(30 + k)
    n_cols = 3 if k <= 14 else 2
    cols = O.random_fr(rng, n_cols * (1 << k)).reshape(n_cols, 1 << k, 4)
    w = O.root_of_unity(k)
    assert np.array_equal(api.root_of_unity(k), w)
    got = api.ntt_batch(cols, w)
    want = O.ntt_batch(cols, w, threads=4)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("k", [3, 10, 13, 16])
def test_lagrange_to_coeff_and_extended(api, O, k):
    rng = np.random.default_rng(50 + k)
    cols = O.random_fr(rng, 2 << k).reshape(2, 1 << k, 4)
    want_c, want_e = O.lde_batch(cols, ext=2, threads=4)
    got_c = api.lagrange_to_coeff(cols)
    assert np.array_equal(got_c, want_c)
    got_e = api.coeff_to_extended(got_c, 2)
    assert np.array_equal(got_e, want_e)


@pytest.mark.parametrize("k,ext", [(7, 3), (9, 1), (10, 1), (10, 3), (12, 1), (13, 3)])
def test_coeff_to_extended_other_extensions(api, O, k, ext):
    """extension factors 2 and 8 (the zero-padded first pass skips stages only when at most a quarter of a row is
    data), single-pass and multi-pass sizes"""
    rng = np.random.default_rng(70 + 10 * k + ext)
    cols = O.random_fr(rng, 2 << k).reshape(2, 1 << k, 4)
    want_c, want_e = O.lde_batch(cols, ext=ext, threads=4)
    got_c = api.lagrange_to_coeff(cols)
    assert np.array_equal(got_c, want_c)
    assert np.array_equal(api.coeff_to_extended(got_c, ext), want_e)


def test_ntt_three_passes_k20(api, O):
    """2^20 = 128 x 128 x 64: the three-pass decomposition (middle pass with two outer digits)"""
    rng = np.random.default_rng(61)
    k = 20
    cols = O.random_fr(rng, 1 << k).reshape(1, 1 << k, 4)
    w = O.root_of_unity(k)
    f = api.ntt_batch(cols, w)
    assert np.array_equal(f, O.ntt_batch(cols, w, threads=4))
    winv = O.fr_inv(w.reshape(1, 4))[0]
    assert np.array_equal(api.ntt_batch(f, winv, api.NTT_INVERSE_SCALE), cols)


def test_ntt_roundtrip_full_size(api, O):
    # size-independent property at the bench size (k=16 -> extended 2^18): iNTT(NTT(x)) == x
    rng = np.random.default_rng(60)
    k = 18
    cols = O.random_fr(rng, 1 << k).reshape(1, 1 << k, 4)
    w = O.root_of_unity(k)
    winv = O.fr_inv(w.reshape(1, 4))[0]
    f = api.ntt_batch(cols, w)
    back = api.ntt_batch(f, winv, api.NTT_INVERSE_SCALE)
    assert np.array_equal(back, cols)
    # linearity spot check against the oracle on a single output coefficient set
    assert np.array_equal(f[0, :4], O.ntt(cols[0], w)[:4])


def test_poseidon(api, O):
    rng = np.random.default_rng(70)
    st = O.fr_from_ints([0, 1, 2]).reshape(1, 3, 4)
    kat = [0x115CC0F5E7D690413DF64C6B9662E9CF2A3617F2743245519E19607A4417189A,
           0x0FCA49B798923AB0239DE1C9E7A4A9A2210312B6A2F616D18B5A87F9B628AE29,
           0x0E7AE82E40091E63CBD4F16A6D16310B3729D4B6E138FCF54110E2867045A30C]
    assert O.fr_to_ints(api.poseidon_permute(st).reshape(3, 4)) == kat
    for ln in (0, 1, 2, 3, 7, 128):
        msgs = O.random_fr(rng, 40 * max(ln, 1)).reshape(40, max(ln, 1), 4)[:, :ln]
        assert np.array_equal(api.poseidon_hash_many(msgs), O.poseidon_hash_many(msgs))
    for n, dim in ((1, 3), (2, 4), (3, 5), (37, 16), (64, 128)):
        v = O.random_fr(rng, n * dim).reshape(n, dim, 4)
        assert np.array_equal(api.poseidon_merkle_root(v), O.poseidon_merkle_root(v))


def test_fr_mul_throughput_report(api):
    rate = api.bench_fr_mul()
    print(f"\nFr Montgomery mul throughput: {rate / 1e9:.1f} G mul/s")
    assert rate > 1e9


@pytest.mark.parametrize("max_s", ["4", "5", "6"])
def test_ntt_small_pass_sizes_in_a_subprocess(max_s):
    """VDB_NTT_MAX_S (read once per process) forces 3- and 4-digit decompositions at small sizes: the general multi-pass
    path (middle passes, digit peeling of the output index, tiles narrower than 1024 elements) against the oracle"""
    import os
    import subprocess
    import sys
    script = (
        "import numpy as np\n"
        "from halo2_vectordb_amd import api\n"
        "from oracle import oracle as O\n"
        "api.init(0)\n"
        "rng = np.random.default_rng(5)\n"
        "for k in (9, 11, 13, 16, 17):\n"
        "    cols = O.random_fr(rng, 2 << k).reshape(2, 1 << k, 4)\n"
        "    w = O.root_of_unity(k)\n"
        "    assert np.array_equal(api.ntt_batch(cols, w), O.ntt_batch(cols, w, threads=4)), k\n"
        "for k in (9, 12):\n"
        "    cols = O.random_fr(rng, 2 << k).reshape(2, 1 << k, 4)\n"
        "    wc, we = O.lde_batch(cols, ext=2, threads=4)\n"
        "    gc = api.lagrange_to_coeff(cols)\n"
        "    assert np.array_equal(gc, wc) and np.array_equal(api.coeff_to_extended(gc, 2), we), k\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VDB_NTT_MAX_S=max_s, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", script], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_lde_k18_maximum_config_size(api, O):
    """BASELINE configs[4] works at 2^18 rows: lagrange_to_coeff (2 passes) and coeff_to_extended to 2^20 (3 passes, the
    zero-padded first pass skipping its two replication stages) against the oracle, one column"""
    rng = np.random.default_rng(1818)
    k = 18
    cols = O.random_fr(rng, 1 << k).reshape(1, 1 << k, 4)
    want_c, want_e = O.lde_batch(cols, ext=2, threads=4)
    got_c = api.lagrange_to_coeff(cols)
    assert np.array_equal(got_c, want_c)
    assert np.array_equal(api.coeff_to_extended(got_c, 2), want_e)


def test_transform_sweep_every_size_and_odd_batches(api, O):
    """every transform size 2^0 .. 2^17 with batch widths that are not a multiple of anything (1 .. 37 columns: the kernels group
    columns per launch and per tile), forward, lagrange_to_coeff and the extension to 2, 4 and 8 times the rows, on random columns
    and on columns with structure (constant, a single one, alternating signs) — against the oracle"""
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    rng = np.random.default_rng(171717)
    for k in range(0, 18):
        n = 1 << k
        n_cols = int(rng.integers(1, 38)) if k <= 12 else int(rng.integers(1, 6))
        cols = O.random_fr(rng, n_cols * n).reshape(n_cols, n, 4)
        cols[0] = O.fr_from_ints([7] * n)                                  # a constant: one non-zero coefficient
        if n_cols > 1:
            cols[1] = 0
            cols[1, n // 3] = O.fr_from_ints([1])[0]                       # a Lagrange basis polynomial
        if n_cols > 2:
            cols[2] = O.fr_from_ints([1, R - 1] * (n // 2) if n > 1 else [1])
        w = O.root_of_unity(k)
        assert np.array_equal(api.ntt_batch(cols, w), O.ntt_batch(cols, w, threads=4)), k
        ext = 1 + k % 3
        want_c, want_e = O.lde_batch(cols, ext=ext, threads=4)
        got_c = api.lagrange_to_coeff(cols)
        assert np.array_equal(got_c, want_c), k
        if k + ext <= 19:
            assert np.array_equal(api.coeff_to_extended(got_c, ext), want_e), (k, ext)


def test_poseidon_sweep_of_message_lengths_and_tree_sizes(api, O):
    """the sponge at every message length 0 .. 21 (an empty absorb, lengths on both sides of every multiple of the rate) with batch sizes
    that fill no wavefront, and merkle_commitment's root for every number of leaves 1 .. 20 (padding to the next power of two at every
    position) at odd and even vector widths — against the oracle"""
    rng = np.random.default_rng(5432)
    for ln in range(0, 22):
        n = int(rng.integers(1, 131))
        msgs = O.random_fr(rng, n * max(ln, 1)).reshape(n, max(ln, 1), 4)[:, :ln]
        assert np.array_equal(api.poseidon_hash_many(msgs), O.poseidon_hash_many(msgs)), (ln, n)
    for n in range(1, 21):
        dim = int(rng.integers(1, 12))
        v = O.random_fr(rng, n * dim).reshape(n, dim, 4)
        assert np.array_equal(api.poseidon_merkle_root(v), O.poseidon_merkle_root(v)), (n, dim)
