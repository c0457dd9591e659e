"""numpy-facing wrappers over the C ABI (include/vdb.h).  Thin plumbing only: every function
marshals numpy buffers into one `vdb_*` call.  Field elements are uint64 arrays (..., 4):
little-endian limbs, Montgomery form (halo2curves layout).  G1 points are (..., 8): x then y.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import VdbError, check  # noqa: F401

NTT_INVERSE_SCALE = 1


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _sz(n):
    return ctypes.c_size_t(int(n))


def _fr(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 4
    return a


def init(device=None):
    return _lib.init(device)


def shutdown():
    _lib.load().vdb_shutdown()
    _lib._inited = None


def device_count():
    return _lib.load().vdb_device_count()


def init_devices(n):
    """Bind GPUs 0 .. n-1 in this process (vdb_init_devices); this thread works on device 0 until set_device."""
    return _lib.init_devices(n)


def set_device(device):
    check(_lib.load().vdb_set_device(int(device)))


def current_device():
    return _lib.load().vdb_current_device()


def devices_bound():
    return _lib.load().vdb_devices_bound()


# ---------------------------------------------------------------- field helpers
def _binop(name, a, b):
    L = _lib.init()
    a, b = _fr(a), _fr(b)
    o = np.empty_like(a)
    check(getattr(L, name)(_p(a), _p(b), _p(o), _sz(a.size // 4)))
    return o


def fr_mul(a, b):
    return _binop("vdb_fr_mul", a, b)


def fr_add(a, b):
    return _binop("vdb_fr_add", a, b)


def fr_sub(a, b):
    return _binop("vdb_fr_sub", a, b)


def _unop(name, a):
    L = _lib.init()
    a = _fr(a)
    o = np.empty_like(a)
    check(getattr(L, name)(_p(a), _p(o), _sz(a.size // 4)))
    return o


def fr_from_canonical(a):
    return _unop("vdb_fr_from_canonical", a)


def fr_to_canonical(a):
    return _unop("vdb_fr_to_canonical", a)


def fr_batch_invert(a):
    return _unop("vdb_fr_batch_invert", a)


def bench_fr_mul(threads=256 * 256 * 8, iters=2000):
    L = _lib.init()
    out = ctypes.c_double(0)
    check(L.vdb_bench_fr_mul(_sz(threads), _sz(iters), ctypes.byref(out)))
    return out.value


def root_of_unity(k):
    o = np.zeros(4, dtype=np.uint64)
    check(_lib.load().vdb_fr_root_of_unity(ctypes.c_uint32(k), _p(o)))
    return o


# ---------------------------------------------------------------- NTT
def _col_ptrs(cols):
    ptrs = (ctypes.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    return ptrs


def ntt_batch(cols, omega, flags=0):
    """cols: (n_cols, n, 4); returns transformed copy (host-pointer entry point)."""
    L = _lib.init()
    cols = np.array(cols, dtype=np.uint64, copy=True)
    n_cols, n = cols.shape[0], cols.shape[1]
    views = [cols[i] for i in range(n_cols)]
    omega = _fr(omega)
    check(L.vdb_ntt_batch(_col_ptrs(views), _sz(n_cols), ctypes.c_uint32(n.bit_length() - 1), _p(omega), ctypes.c_int(flags)))
    return cols


def lagrange_to_coeff(cols):
    L = _lib.init()
    cols = np.array(cols, dtype=np.uint64, copy=True)
    n_cols, n = cols.shape[0], cols.shape[1]
    views = [cols[i] for i in range(n_cols)]
    check(L.vdb_lagrange_to_coeff(_col_ptrs(views), _sz(n_cols), ctypes.c_uint32(n.bit_length() - 1)))
    return cols


def coeff_to_extended(cols, ext_k=2):
    L = _lib.init()
    cols = _fr(cols)
    n_cols, n = cols.shape[0], cols.shape[1]
    out = np.zeros((n_cols, n << ext_k, 4), dtype=np.uint64)
    check(L.vdb_coeff_to_extended(_col_ptrs([cols[i] for i in range(n_cols)]), _col_ptrs([out[i] for i in range(n_cols)]),
                                  _sz(n_cols), ctypes.c_uint32(n.bit_length() - 1), ctypes.c_uint32(ext_k)))
    return out


# ---------------------------------------------------------------- SRS / MSM
class Srs:
    """Device-resident KZG bases + fixed-base window tables (vdb_srs_load)."""

    def __init__(self, k, g=None, g_lagrange=None, window_bits=0):
        """window_bits: Pippenger window (0 = default, tuned for witness columns; 14 suits columns of full-width scalars)"""
        self.L = _lib.init()
        self.k = k
        h = ctypes.c_void_p()
        ga = np.ascontiguousarray(g, dtype=np.uint64) if g is not None else None
        gl = np.ascontiguousarray(g_lagrange, dtype=np.uint64) if g_lagrange is not None else None
        check(self.L.vdb_srs_load_window(ctypes.c_uint32(k), _p(ga) if ga is not None else None, _p(gl) if gl is not None else None,
                                         ctypes.c_uint32(window_bits), ctypes.byref(h)))
        self.h = h

    def info(self):
        k, c, w = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        check(self.L.vdb_srs_info(self.h, ctypes.byref(k), ctypes.byref(c), ctypes.byref(w)))
        return k.value, c.value, w.value

    def free(self):
        if self.h is not None:
            self.L.vdb_srs_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class SrsAll:
    """One Srs handle per bound device, the bases replicated (vdb_srs_load_all); commits batches of columns over all of them."""

    def __init__(self, k, g=None, g_lagrange=None, window_bits=0):
        self.L = _lib.init()
        self.k = k
        n = devices_bound()
        self.handles = (ctypes.c_void_p * n)()
        ga = np.ascontiguousarray(g, dtype=np.uint64) if g is not None else None
        gl = np.ascontiguousarray(g_lagrange, dtype=np.uint64) if g_lagrange is not None else None
        check(self.L.vdb_srs_load_all(ctypes.c_uint32(k), _p(ga) if ga is not None else None, _p(gl) if gl is not None else None,
                                      ctypes.c_uint32(window_bits), self.handles, n))

    def devices(self):
        out = []
        for h in self.handles:
            d = ctypes.c_int(-1)
            check(self.L.vdb_srs_device(h, ctypes.byref(d)))
            out.append(d.value)
        return out

    def msm_batch(self, cols, basis=1):
        """cols: (n_cols, n, 4) host scalars -> (n_cols, 8): device i commits the i-th contiguous block of the columns"""
        cols = _fr(cols)
        n_cols, n = cols.shape[0], cols.shape[1]
        out = np.zeros((n_cols, 8), dtype=np.uint64)
        check(self.L.vdb_msm_batch_multi(self.handles, len(self.handles), ctypes.c_int(basis), _col_ptrs([cols[i] for i in range(n_cols)]),
                                         _sz(n_cols), _sz(n), _p(out)))
        return out

    def free(self):
        for i, h in enumerate(self.handles):
            if h:
                self.L.vdb_srs_free(h)
                self.handles[i] = None


def ntt_batch_multi(cols, omega, flags=0):
    """ntt_batch with the columns cut into one block per bound device"""
    L = _lib.init()
    cols = np.array(cols, dtype=np.uint64, copy=True)
    n_cols, n = cols.shape[0], cols.shape[1]
    omega = _fr(omega)
    check(L.vdb_ntt_batch_multi(_col_ptrs([cols[i] for i in range(n_cols)]), _sz(n_cols), ctypes.c_uint32(n.bit_length() - 1), _p(omega), ctypes.c_int(flags)))
    return cols


def srs_setup_unsafe(k, tau_mont):
    """(g, g_lagrange) for a known tau (Montgomery Fr limbs)."""
    lib = _lib.init()
    tau = _fr(tau_mont)
    g = np.zeros((1 << k, 8), dtype=np.uint64)
    gl = np.zeros((1 << k, 8), dtype=np.uint64)
    check(lib.vdb_srs_setup_unsafe(ctypes.c_uint32(k), _p(tau), _p(g), _p(gl)))
    return g, gl


def msm_batch(srs, cols, basis=1):
    """cols: (n_cols, n, 4) Montgomery scalars (host) -> (n_cols, 8) affine points."""
    cols = _fr(cols)
    n_cols, n = cols.shape[0], cols.shape[1]
    out = np.zeros((n_cols, 8), dtype=np.uint64)
    check(srs.L.vdb_msm_batch(srs.h, ctypes.c_int(basis), _col_ptrs([cols[i] for i in range(n_cols)]), _sz(n_cols), _sz(n), _p(out)))
    return out


def msm(srs, scalars, basis=1):
    scalars = _fr(scalars)
    out = np.zeros(8, dtype=np.uint64)
    check(srs.L.vdb_msm(srs.h, ctypes.c_int(basis), _p(scalars), _sz(scalars.size // 4), _p(out)))
    return out


def msm_batch_dev(srs, dev_ptr, n_cols, n, basis=1):
    out = np.zeros((n_cols, 8), dtype=np.uint64)
    check(srs.L.vdb_msm_batch_dev(srs.h, ctypes.c_int(basis), dev_ptr, _sz(n_cols), _sz(n), _p(out)))
    return out


# ---------------------------------------------------------------- fixed point + witness streams
METRICS = dict(euclidean=0, cosine=1, manhattan=2, hamming=3)


def quantize(x, P=48):
    x = np.ascontiguousarray(x, dtype=np.float64)
    o = np.zeros(x.shape + (4,), dtype=np.uint64)
    check(_lib.load().vdb_fp_quantize(ctypes.c_uint32(P), _p(x), _p(o), _sz(x.size)))
    return o


def dequantize(a, P=48):
    a = _fr(a)
    o = np.zeros(a.shape[:-1], dtype=np.float64)
    check(_lib.load().vdb_fp_dequantize(ctypes.c_uint32(P), _p(a), _p(o), _sz(a.size // 4)))
    return o


def _u64():
    return ctypes.c_uint64(0)


def _split_flags(flags):
    """flag byte per advice cell: bit 0 = gate start (selector), bit 1 = data-independent constant cell"""
    if flags is None:
        return dict(selectors=None, const_mask=None, flags=None)
    return dict(selectors=flags & 1, const_mask=(flags >> 1) & 1, flags=flags)


def wit_distance(metric, a, b, P=48, L=13, selectors=False):
    """a, b: (n_pairs, dim, 4).  Returns dict(stream, lookup, selectors, result)."""
    lib = _lib.init()
    a, b = _fr(a), _fr(b)
    n, dim = a.shape[0], a.shape[1]
    cells, lk = _u64(), _u64()
    check(lib.vdb_wit_distance_size(METRICS[metric], ctypes.c_uint32(P), ctypes.c_uint32(L), _sz(n), _sz(dim), ctypes.byref(cells), ctypes.byref(lk)))
    stream = np.zeros((cells.value, 4), dtype=np.uint64)
    lookup = np.zeros((lk.value, 4), dtype=np.uint64)
    sel = np.zeros(cells.value, dtype=np.uint8) if selectors else None
    res = np.zeros((n, 4), dtype=np.uint64)
    check(lib.vdb_wit_distance(METRICS[metric], ctypes.c_uint32(P), ctypes.c_uint32(L), _p(a), _p(b), _sz(n), _sz(dim), _p(stream), _p(lookup),
                               _p(sel) if selectors else None, _p(res)))
    return dict(stream=stream, lookup=lookup, **_split_flags(sel), result=res)


FP_OPS = dict(qadd=0, qsub=1, qmul=2, qdiv=3, neg=4, qabs=5, is_neg=6, qmin=7, qsqrt=8, qlog2=9, qexp2=10, qlog=11, qexp=12, qpow=13, bit_xor=14,
              cond_neg=15, signed_div_scale=16, qmax=17, sign=18, clip=19, qmod=20, qsin=21, qcos=22, qtan=23, qsinh=24, qcosh=25, qtanh=26)


def wit_fp_op(op, a, b=None, P=48, L=13, selectors=False):
    """n independent FixedPointChip calls op(a_i [, b_i]) (vdb_wit_fp_op): a, b (n, 4) quantized values; instance i's cells are
    stream[i * cells_per_call : (i + 1) * cells_per_call]"""
    lib = _lib.init()
    a = _fr(a).reshape(-1, 4)
    n = a.shape[0]
    b = None if b is None else _fr(b).reshape(-1, 4)
    assert b is None or b.shape == a.shape
    cells, lk = _u64(), _u64()
    check(lib.vdb_wit_fp_op_size(FP_OPS[op], ctypes.c_uint32(P), ctypes.c_uint32(L), _sz(n), ctypes.byref(cells), ctypes.byref(lk)))
    stream = np.zeros((cells.value, 4), dtype=np.uint64)
    lookup = np.zeros((lk.value, 4), dtype=np.uint64)
    sel = np.zeros(cells.value, dtype=np.uint8) if selectors else None
    res = np.zeros((n, 4), dtype=np.uint64)
    check(lib.vdb_wit_fp_op(FP_OPS[op], ctypes.c_uint32(P), ctypes.c_uint32(L), _p(a), _p(b) if b is not None else None, _sz(n), _p(stream), _p(lookup),
                            _p(sel) if selectors else None, _p(res)))
    return dict(stream=stream, lookup=lookup, **_split_flags(sel), result=res)


def wit_nearest(metric, query, vectors, P=48, L=13, selectors=False):
    lib = _lib.init()
    query, vectors = _fr(query), _fr(vectors)
    n, dim = vectors.shape[0], vectors.shape[1]
    cells, lk = _u64(), _u64()
    check(lib.vdb_wit_nearest_size(METRICS[metric], ctypes.c_uint32(P), ctypes.c_uint32(L), _sz(n), _sz(dim), ctypes.byref(cells), ctypes.byref(lk)))
    stream = np.zeros((cells.value, 4), dtype=np.uint64)
    lookup = np.zeros((lk.value, 4), dtype=np.uint64)
    sel = np.zeros(cells.value, dtype=np.uint8) if selectors else None
    ind = np.zeros((n, 4), dtype=np.uint64)
    res = np.zeros((dim, 4), dtype=np.uint64)
    check(lib.vdb_wit_nearest(METRICS[metric], ctypes.c_uint32(P), ctypes.c_uint32(L), _p(query), _p(vectors), _sz(n), _sz(dim), _p(stream), _p(lookup),
                              _p(sel) if selectors else None, _p(ind), _p(res)))
    return dict(stream=stream, lookup=lookup, **_split_flags(sel), indicator=ind, result=res)


def wit_kmeans(metric, vectors, K, I, P=48, L=13, zero_cached=False, selectors=False):
    lib = _lib.init()
    vectors = _fr(vectors)
    n, dim = vectors.shape[0], vectors.shape[1]
    cells, lk = _u64(), _u64()
    check(lib.vdb_wit_kmeans_size(METRICS[metric], ctypes.c_uint32(P), ctypes.c_uint32(L), _sz(n), _sz(dim), _sz(K), _sz(I), int(zero_cached),
                                  ctypes.byref(cells), ctypes.byref(lk)))
    stream = np.zeros((cells.value, 4), dtype=np.uint64)
    lookup = np.zeros((lk.value, 4), dtype=np.uint64)
    sel = np.zeros(cells.value, dtype=np.uint8) if selectors else None
    cent = np.zeros((K, dim, 4), dtype=np.uint64)
    ind = np.zeros((n, K, 4), dtype=np.uint64)
    check(lib.vdb_wit_kmeans(METRICS[metric], ctypes.c_uint32(P), ctypes.c_uint32(L), _p(vectors), _sz(n), _sz(dim), _sz(K), _sz(I), int(zero_cached),
                             _p(stream), _p(lookup), _p(sel) if selectors else None, _p(cent), _p(ind)))
    return dict(stream=stream, lookup=lookup, **_split_flags(sel), centroids=cent, indicators=ind)


def wit_merkle(vectors, zero_cached=False, selectors=False):
    lib = _lib.init()
    vectors = _fr(vectors)
    n, dim = vectors.shape[0], vectors.shape[1]
    cells = _u64()
    check(lib.vdb_wit_merkle_size(_sz(n), _sz(dim), int(zero_cached), ctypes.byref(cells)))
    stream = np.zeros((cells.value, 4), dtype=np.uint64)
    sel = np.zeros(cells.value, dtype=np.uint8) if selectors else None
    root = np.zeros(4, dtype=np.uint64)
    check(lib.vdb_wit_merkle(_p(vectors), _sz(n), _sz(dim), int(zero_cached), _p(stream), _p(sel) if selectors else None, _p(root)))
    return dict(stream=stream, **_split_flags(sel), root=root)


def layout_plan(selectors, k, minimum_rows=9):
    lib = _lib.init()
    selectors = np.ascontiguousarray(selectors, dtype=np.uint8)
    nbp = _u64()
    check(lib.vdb_layout_plan(_p(selectors), ctypes.c_uint64(selectors.size), ctypes.c_uint32(k), ctypes.c_uint32(minimum_rows), None, ctypes.c_uint64(0),
                              ctypes.byref(nbp)))
    bp = np.zeros(max(nbp.value, 1), dtype=np.uint64)
    check(lib.vdb_layout_plan(_p(selectors), ctypes.c_uint64(selectors.size), ctypes.c_uint32(k), ctypes.c_uint32(minimum_rows), _p(bp),
                              ctypes.c_uint64(bp.size), ctypes.byref(nbp)))
    return bp[:nbp.value]


def layout_columns(stream, break_points, k, lookup=None, minimum_rows=9):
    lib = _lib.init()
    stream = _fr(stream)
    bp = np.ascontiguousarray(break_points, dtype=np.uint64)
    cols = np.zeros((bp.size + 1, 1 << k, 4), dtype=np.uint64)
    lcols = None
    n_lc = 0
    if lookup is not None and len(lookup):
        lookup = _fr(lookup)
        max_rows = (1 << k) - minimum_rows
        n_lc = (lookup.shape[0] + max_rows - 1) // max_rows
        lcols = np.zeros((n_lc, 1 << k, 4), dtype=np.uint64)
    check(lib.vdb_layout_columns(_p(stream), ctypes.c_uint64(stream.shape[0]), _p(bp), ctypes.c_uint64(bp.size), _p(lookup) if lcols is not None else None,
                                 ctypes.c_uint64(lookup.shape[0] if lcols is not None else 0), ctypes.c_uint32(k), ctypes.c_uint32(minimum_rows), _p(cols),
                                 _p(lcols) if lcols is not None else None, ctypes.c_uint64(n_lc)))
    return cols, lcols


# ---------------------------------------------------------------- Poseidon
def poseidon_hash_many(msgs):
    L = _lib.init()
    msgs = _fr(msgs)
    n, ln = msgs.shape[0], msgs.shape[1]
    o = np.zeros((n, 4), dtype=np.uint64)
    check(L.vdb_poseidon_hash_many(_p(msgs), _sz(n), _sz(ln), _p(o)))
    return o


def poseidon_merkle_root(vectors):
    L = _lib.init()
    vectors = _fr(vectors)
    o = np.zeros(4, dtype=np.uint64)
    check(L.vdb_poseidon_merkle_root(_p(vectors), _sz(vectors.shape[0]), _sz(vectors.shape[1]), _p(o)))
    return o


def poseidon_permute(states):
    L = _lib.init()
    s = np.array(states, dtype=np.uint64, copy=True)
    check(L.vdb_poseidon_permute(_p(s), _sz(s.size // 12)))
    return s


def random_scalars_dev(dst_ptr, n, seed=None):
    """n uniformly random field elements written to device memory at dst_ptr: 64 bytes of entropy each, reduced mod r on
    the device (vdb_fr_from_wide_dev = halo2curves Fr::random).  seed=None draws from the operating system (os.urandom), as
    halo2's prover does for its blinding scalars; an integer seed gives a reproducible stream (tests only)."""
    import os
    lib = _lib.init()
    raw = os.urandom(64 * n) if seed is None else np.random.default_rng(seed).bytes(64 * n)
    wide = DeviceBuffer(max(64 * n, 64))
    try:
        wide.upload(np.frombuffer(raw, dtype=np.uint8))
        check(lib.vdb_fr_from_wide_dev(wide.ptr, _sz(n), dst_ptr))
        sync()        # `wide` is freed on return
    finally:
        wide.free()


class MockReport(ctypes.Structure):
    """vdb_mock_report (include/vdb.h)"""
    _fields_ = [(n, ctypes.c_uint64) for n in ("gate_rows_violated", "first_gate_row", "lookup_cells_out_of_table", "first_lookup_cell",
                                               "copies_unequal", "first_copy", "lookup_copies_unequal", "first_lookup_copy",
                                               "constants_changed", "first_constant", "instances_unequal", "first_instance")]

    def violations(self):
        return int(self.gate_rows_violated + self.lookup_cells_out_of_table + self.copies_unequal + self.lookup_copies_unequal + self.constants_changed
                   + self.instances_unequal)

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def mock_check_dev(stream_ptr, n_cells, flags_ptr, lookup_ptr, n_lookup, lookup_bits, copy_of_ptr=None, lookup_src_ptr=None, const_stream_ptr=None,
                   const_idx_ptr=None, const_table_ptr=None, n_consts=0):
    """MockProver-style check of a device-resident witness (vdb_mock_check_dev): every gate row, lookup cell, copy and constant"""
    rep = MockReport()
    check(_lib.init().vdb_mock_check_dev(stream_ptr, ctypes.c_uint64(n_cells), flags_ptr, lookup_ptr, ctypes.c_uint64(n_lookup), ctypes.c_uint32(lookup_bits),
                                         copy_of_ptr, lookup_src_ptr, const_stream_ptr, const_idx_ptr, const_table_ptr, ctypes.c_uint64(n_consts),
                                         ctypes.byref(rep)))
    return rep


def mock_check_instances_dev(rep, stream_ptr, n_cells, cells_ptr, values_ptr, n_instances):
    """the `instances` argument of MockProver::run on a device-resident witness (vdb_mock_check_instances_dev): stream cell cells[i]
    must hold values[i]; fills rep.instances_unequal / first_instance"""
    check(_lib.init().vdb_mock_check_instances_dev(stream_ptr, ctypes.c_uint64(n_cells), cells_ptr, values_ptr, ctypes.c_uint64(n_instances), ctypes.byref(rep)))
    return rep


def mock_check(stream, flags, lookup, lookup_bits, copy_of=None, lookup_src=None, const_stream=None):
    """the same on host arrays (uploaded): stream (n, 4), flags (n,) uint8, lookup (m, 4); copy maps as int64 arrays"""
    stream, lookup = _fr(stream), _fr(lookup) if lookup is not None and len(lookup) else np.zeros((0, 4), dtype=np.uint64)
    arrays = [stream, np.ascontiguousarray(flags, dtype=np.uint8), lookup]
    opt = [None if a is None else np.ascontiguousarray(a, dtype=np.int64) for a in (copy_of, lookup_src)] + [None if const_stream is None else _fr(const_stream)]
    bufs = [DeviceBuffer(max(a.nbytes, 32)) for a in arrays] + [None if a is None else DeviceBuffer(max(a.nbytes, 32)) for a in opt]
    try:
        for b, a in zip(bufs, arrays + opt):
            if b is not None and a.nbytes:
                b.upload(a)
        return mock_check_dev(bufs[0].ptr, stream.shape[0], bufs[1].ptr, bufs[2].ptr, lookup.shape[0], lookup_bits,
                              *(None if b is None else b.ptr for b in bufs[3:]))
    finally:
        for b in bufs:
            if b is not None:
                b.free()


# ---------------------------------------------------------------- device buffers
def extended_to_coeff(ext_cols, k, ext_k=2):
    """inverse of coeff_to_extended; ext_cols: (n_cols, 2^(k+ext_k), 4) on the host"""
    lib = _lib.init()
    e = _fr(ext_cols)
    buf = DeviceBuffer(max(e.nbytes, 32))
    try:
        buf.upload(e)
        check(lib.vdb_extended_to_coeff_dev(buf.ptr, _sz(e.shape[0]), ctypes.c_uint32(k), ctypes.c_uint32(ext_k)))
        sync()
        return buf.download(e.shape)
    finally:
        buf.free()


def eval_polys(coeffs, x):
    """out[c] = sum_i coeffs[c][i] * x^i; coeffs: (n_cols, n, 4) uint64 on the host (uploaded), x: (4,)"""
    lib = _lib.init()
    coeffs, x = _fr(coeffs), _fr(x)
    out = np.zeros((coeffs.shape[0], 4), dtype=np.uint64)
    buf = DeviceBuffer(max(coeffs.nbytes, 32))
    try:
        buf.upload(coeffs)
        check(lib.vdb_eval_polys_dev(buf.ptr, _sz(coeffs.shape[0]), _sz(coeffs.shape[1]), _p(x), _p(out)))
        return out
    finally:
        buf.free()


def lookup_permute(inputs, table, usable_rows, max_bits):
    """inputs: (n_cols, n, 4), table: (n, 4) Montgomery Fr on the host; returns (permuted_input, permuted_table)"""
    lib = _lib.init()
    inputs, table = _fr(inputs), _fr(table)
    n_cols, n = inputs.shape[0], inputs.shape[1]
    bufs = [DeviceBuffer(max(inputs.nbytes, 32)), DeviceBuffer(max(table.nbytes, 32)), DeviceBuffer(max(inputs.nbytes, 32)), DeviceBuffer(max(inputs.nbytes, 32))]
    try:
        bufs[0].upload(inputs)
        bufs[1].upload(table)
        check(lib.vdb_lookup_permute_dev(bufs[0].ptr, bufs[1].ptr, _sz(n_cols), _sz(n), _sz(usable_rows), ctypes.c_uint32(max_bits), bufs[2].ptr, bufs[3].ptr))
        sync()
        return bufs[2].download((n_cols, n, 4)), bufs[3].download((n_cols, n, 4))
    finally:
        for b in bufs:
            b.free()


def fr_delta():
    """halo2curves bn256 Fr::DELTA (Montgomery)"""
    out = np.zeros(4, dtype=np.uint64)
    check(_lib.load().vdb_fr_delta(_p(out)))
    return out


def _with_buffers(arrays, n_out_bytes, call):
    bufs = [DeviceBuffer(max(a.nbytes, 32)) for a in arrays] + [DeviceBuffer(max(nb, 32)) for nb in n_out_bytes]
    try:
        for b, a in zip(bufs, arrays):
            b.upload(a)
        call(*[b.ptr for b in bufs])
        sync()
        return bufs[len(arrays):]
    except Exception:
        for b in bufs[len(arrays):]:
            b.free()
        raise
    finally:
        for b in bufs[: len(arrays)]:
            b.free()


def permutation_sigma(mapping, k, delta=None):
    """mapping: (n_cols, 2^k) uint64 words col' << 32 | row'; returns the sigma columns (n_cols, 2^k, 4)"""
    lib = _lib.init()
    mapping = np.ascontiguousarray(mapping, dtype=np.uint64)
    n_cols, n = mapping.shape
    delta = fr_delta() if delta is None else _fr(delta)
    (out,) = _with_buffers([mapping], [n_cols * n * 32],
                           lambda m, o: check(lib.vdb_permutation_sigma_dev(m, _sz(n_cols), ctypes.c_uint32(k), _p(delta), o)))
    try:
        return out.download((n_cols, n, 4))
    finally:
        out.free()


def permutation_product(cols, sigma, usable_rows, chunk_len, beta, gamma, delta=None):
    """running products of the permutation argument, one column per chunk of chunk_len columns: (n_chunks, 2^k, 4)"""
    lib = _lib.init()
    cols, sigma = _fr(cols), _fr(sigma)
    n_cols, n = cols.shape[0], cols.shape[1]
    k = n.bit_length() - 1
    n_chunks = -(-n_cols // chunk_len)
    delta = fr_delta() if delta is None else _fr(delta)
    beta, gamma = _fr(beta), _fr(gamma)
    (out,) = _with_buffers([cols, sigma], [n_chunks * n * 32], lambda c, s, o: check(lib.vdb_permutation_product_dev(
        c, s, _sz(n_cols), ctypes.c_uint32(k), _sz(usable_rows), _sz(chunk_len), _p(beta), _p(gamma), _p(delta), o)))
    try:
        return out.download((n_chunks, n, 4))
    finally:
        out.free()


def lookup_product(inputs, table, perm_inputs, perm_table, usable_rows, beta, gamma):
    """running products of the lookup argument, one per input column: (n_cols, n, 4)"""
    lib = _lib.init()
    inputs, table, perm_inputs, perm_table = _fr(inputs), _fr(table), _fr(perm_inputs), _fr(perm_table)
    n_cols, n = inputs.shape[0], inputs.shape[1]
    beta, gamma = _fr(beta), _fr(gamma)
    (out,) = _with_buffers([inputs, table, perm_inputs, perm_table], [n_cols * n * 32], lambda a, t, pa, pt, o: check(lib.vdb_lookup_product_dev(
        a, t, pa, pt, _sz(n_cols), _sz(n), _sz(usable_rows), _p(beta), _p(gamma), o)))
    try:
        return out.download((n_cols, n, 4))
    finally:
        out.free()


def kate_div(coeffs, x):
    """(p(X) - p(x)) / (X - x) per polynomial; returns (quotients (n_cols, n, 4), remainders p(x) (n_cols, 4))"""
    lib = _lib.init()
    coeffs, x = _fr(coeffs), _fr(x)
    n_cols, n = coeffs.shape[0], coeffs.shape[1]
    rem = np.zeros((n_cols, 4), dtype=np.uint64)
    (out,) = _with_buffers([coeffs], [n_cols * n * 32], lambda c, o: check(lib.vdb_kate_div_dev(c, _sz(n_cols), _sz(n), _p(x), o, _p(rem))))
    try:
        return out.download((n_cols, n, 4)), rem
    finally:
        out.free()


def poly_lincomb(polys, v):
    """sum_c v^(n_cols-1-c) p_c, coefficient-wise: (n, 4)"""
    lib = _lib.init()
    polys, v = _fr(polys), _fr(v)
    n_cols, n = polys.shape[0], polys.shape[1]
    zero = np.zeros((n, 4), dtype=np.uint64)
    bufs = [DeviceBuffer(max(polys.nbytes, 32)), DeviceBuffer(max(zero.nbytes, 32))]
    try:
        bufs[0].upload(polys)
        bufs[1].upload(zero)
        check(lib.vdb_poly_lincomb_dev(bufs[0].ptr, _sz(n_cols), _sz(n), _p(v), bufs[1].ptr))
        sync()
        return bufs[1].download((n, 4))
    finally:
        for b in bufs:
            b.free()


def grand_product(num, den):
    """z[c][0] = 1, z[c][i+1] = z[c][i] * num[c][i] / den[c][i]; num, den: (n_cols, n, 4) uint64 (Montgomery Fr)."""
    lib = _lib.init()
    num, den = _fr(num), _fr(den)
    n_cols, n = num.shape[0], num.shape[1]
    bufs = [DeviceBuffer(max(num.nbytes, 32)) for _ in range(3)]
    try:
        bufs[0].upload(num)
        bufs[1].upload(den)
        check(lib.vdb_grand_product_dev(bufs[0].ptr, bufs[1].ptr, _sz(n_cols), _sz(n), bufs[2].ptr))
        sync()
        return bufs[2].download((n_cols, n, 4))
    finally:
        for b in bufs:
            b.free()


class Transcript:
    """Fiat–Shamir transcript + proof bytes (vdb_transcript_*): Poseidon sponge of width t; host code, no GPU needed."""

    def __init__(self, t=5, r_f=8, r_p=60):
        self.L = _lib.load()
        self.h = ctypes.c_void_p()
        check(self.L.vdb_transcript_new(t, r_f, r_p, ctypes.byref(self.h)))

    def set_sign_bit(self, bit):
        """bit of a compressed point's last byte that says "y is odd": 6 (default) or 7"""
        check(self.L.vdb_transcript_set_sign_bit(self.h, ctypes.c_uint32(bit)))

    def common_scalar(self, s):
        check(self.L.vdb_transcript_common_scalar(self.h, _p(_fr(s))))

    def common_point(self, pt):
        check(self.L.vdb_transcript_common_point(self.h, _p(np.ascontiguousarray(pt, dtype=np.uint64))))

    def write_scalar(self, s):
        check(self.L.vdb_transcript_write_scalar(self.h, _p(_fr(s))))

    def write_point(self, pt):
        check(self.L.vdb_transcript_write_point(self.h, _p(np.ascontiguousarray(pt, dtype=np.uint64))))

    def write_points(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 8)
        check(self.L.vdb_transcript_write_points(self.h, _p(pts), _sz(pts.shape[0])))

    def write_scalars(self, ss):
        ss = np.ascontiguousarray(ss, dtype=np.uint64).reshape(-1, 4)
        check(self.L.vdb_transcript_write_scalars(self.h, _p(ss), _sz(ss.shape[0])))

    def common_points(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 8)
        check(self.L.vdb_transcript_common_points(self.h, _p(pts), _sz(pts.shape[0])))

    def common_scalars(self, ss):
        ss = np.ascontiguousarray(ss, dtype=np.uint64).reshape(-1, 4)
        check(self.L.vdb_transcript_common_scalars(self.h, _p(ss), _sz(ss.shape[0])))

    def flush(self):
        """absorb the complete chunks written so far now (vdb_transcript_flush): host work beside queued device work"""
        check(self.L.vdb_transcript_flush(self.h))

    def squeeze(self):
        out = np.zeros(4, dtype=np.uint64)
        check(self.L.vdb_transcript_squeeze(self.h, _p(out)))
        return out

    def proof(self):
        n = ctypes.c_size_t()
        check(self.L.vdb_transcript_proof_len(self.h, ctypes.byref(n)))
        buf = np.zeros(max(n.value, 1), dtype=np.uint8)
        check(self.L.vdb_transcript_proof_bytes(self.h, _p(buf), _sz(buf.size)))
        return buf[: n.value].tobytes()

    def free(self):
        if self.h:
            self.L.vdb_transcript_free(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """A raw HBM allocation owned by the library's context."""

    def __init__(self, nbytes):
        self.L = _lib.init()
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(self.L.vdb_malloc(ctypes.byref(p), _sz(self.nbytes)))
        self.ptr = p

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        check(self.L.vdb_memcpy_h2d(ctypes.c_void_p(self.ptr.value + offset), _p(arr), _sz(arr.nbytes)))

    def download(self, shape, dtype=np.uint64, offset=0):
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        check(self.L.vdb_memcpy_d2h(_p(out), ctypes.c_void_p(self.ptr.value + offset), _sz(out.nbytes)))
        return out

    def at(self, offset):
        return ctypes.c_void_p(self.ptr.value + int(offset))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            check(self.L.vdb_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def mem_info():
    """(free, total) bytes of HBM on this device"""
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    check(_lib.init().vdb_mem_info(ctypes.byref(f), ctypes.byref(t)))
    return f.value, t.value


def scratch_held():
    """bytes the library's cached work buffers hold (what vdb_scratch_release would give back)"""
    b = ctypes.c_size_t()
    check(_lib.init().vdb_scratch_held(ctypes.byref(b)))
    return b.value


def alloc_stats(reset=False):
    """(seconds, bytes, calls) of the device allocations made through the library since start / the last reset (vdb_alloc_stats)"""
    sec, b, n = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
    _lib.load().vdb_alloc_stats(ctypes.byref(sec), ctypes.byref(b), ctypes.byref(n), 1 if reset else 0)
    return sec.value, b.value, n.value


KEYGEN_SCRATCH_CAP = 16 << 30      # bound of the MSM's work space while setup / keygen allocate the proving key (vdb_msm_set_scratch_cap)


def msm_scratch_cap(nbytes):
    check(_lib.init().vdb_msm_set_scratch_cap(_sz(nbytes)))


def sync():
    check(_lib.init().vdb_sync())


def timer_start():
    check(_lib.init().vdb_timer_start())


def timer_stop():
    ms = ctypes.c_float(0)
    check(_lib.init().vdb_timer_stop(ctypes.byref(ms)))
    return ms.value


def profile_begin(deferred=False):
    """deferred: do not wait for each launch; the events are read in profile_end (kernels run as in the untimed path)"""
    check(_lib.init().vdb_profile_begin_deferred() if deferred else _lib.init().vdb_profile_begin())


def profile_end():
    import json
    buf = ctypes.create_string_buffer(1 << 16)
    check(_lib.init().vdb_profile_end(buf, _sz(len(buf))))
    return json.loads(buf.value.decode())
