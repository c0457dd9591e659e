"""The prover's rounds after the advice commitments, composed from the device bricks on the buffers the
k-means hot path leaves resident: lookup permutation -> products -> quotient -> evaluations -> openings, every polynomial
step on the GPU, nothing but commitments, evaluations and challenges crossing the C ABI.

The circuit proved is the reference's circuit: the vertical gate on every gate row, every lookup cell in the range table,
and the permutation argument over the advice columns, the lookup columns, ONE fixed column of constants and ONE instance
column, with every copy halo2-base records while the closure runs — `Existing` cells, `Constant` cells (tied to the fixed
column), constrain_equal / assert_is_const, the lookup cells (copies of advice cells), the overlap cell the column layout
duplicates — taken from the circuit's symbolic map (circuit_sym.py for the fixed-point gadgets, copymap.py for the Merkle
circuit), and every cell the closure pushes into `make_public` tied to its row of the instance column
(RangeWithInstanceCircuitBuilder, /root/reference/src/scaffold/mod.rs:400; the values are circuit.instances(), :265): the
centroids for k-means (examples/kmeans.rs:51-56), the result vector for nearest_vector (examples/query.rs:58), the root for
the Merkle circuit (examples/merkle.rs:47).  The instance polynomial is not committed: prover and verifier both make it from
the public values, which enter the transcript first (halo2's KZG provers: QUERY_INSTANCE = false, [UPSTREAM-RECALL]).
What this is not: halo2's exact proof layout (the order of terms and challenges is recalled, [UPSTREAM-RECALL]; the
Fiat–Shamir transcript is the library's own, vdb_transcript_*; the multi-open is SHPLONK, or one quotient per rotation point
with multiopen="gwc"); the order of the quotient's terms follows plonk/evaluation.rs as recalled ([UPSTREAM-RECALL]; parity
unpinned).  What the tests hold it to instead is what a verifier checks: the quotient identity at a random point recombined
from the returned evaluations, and every opening against its commitments in the exponent (tests/test_gpu_rounds.py).
"""
import ctypes
from collections.abc import Mapping

import os
import time

import numpy as np

from . import api
from ._lib import check
from .pipeline import N_BLIND

B = 32
DERIVED = ("hf",)   # opened polynomials whose evaluation is not in the proof: the verifier computes it (h folded at x, from the quotient identity)
FIXED = ("sel", "sigma", "cst", "table")     # the committed fixed polynomials, in the order the verifying key's digest absorbs their commitments
# (the Lagrange selectors l_0, l_last, l_active = 1 - l_last - l_blind the quotient multiplies by are not polynomials of the key: halo2
#  neither commits nor opens them, its verifier evaluates them at x from the domain — lagrange_evals below; the prover keeps their
#  cosets as key material, self.fixed["lag"])
# The constraint system's degree, halo2 ConstraintSystem::degree() [UPSTREAM-RECALL; SURVEY App. C.4 / C.5]: the maximum of the
# permutation argument's required degree (3), the lookup arguments' (max(4, 2 + input degree + table degree) = 4: halo2-base's "lookup wo
# selector" reads one lookup-advice column against the table column, both of degree 1) and the gates' (the vertical gate
# q (a + b c - d): 3).  A circuit with lookup columns has degree 4; one without (merkle_commitment alone: the builder's auto-config
# gives it no lookup-advice column, so RangeConfig creates no lookup argument) degree 3.  Everything below follows from it the way
# halo2 derives it:
#   chunk_len = degree - 2   columns per product polynomial of the permutation argument (permutation::Argument)
#   n_h       = degree - 1   pieces of the quotient (quotient_poly_degree; extended domain 2^(k + ceil(log2(degree - 1))))
# halo2 evaluates the numerator on every point of the extended domain; the quotient has degree below n_h n, so its values on n_h of the
# cosets of the 2^k-th roots of unity determine it, and the gates' share (degree 3: below 2 n) on two.  The rounds work coset by coset
# ("slots", vdb_coeff_to_cosets_dev: arrays [column][slot][row]) on n_slots = n_h of them and never make the rest: with degree 4 a
# quarter of every extended transform and of every evaluation kernel is not run, and the quotient that comes out is the same polynomial.
def constraint_degree(n_lookup_columns):
    return 4 if n_lookup_columns else 3


GATE_SLOTS = 2               # cosets the gate terms are evaluated on (slots 0, 1 = the coset of 2 n points)
BLOCK_COLS = 510 # fixed-polynomial cosets are produced this many columns at a time (a multiple of every chunk_len; 3.2 GB at 2^16 rows)
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _sz(v):
    return ctypes.c_size_t(int(v))


def _fr_from_int(v):
    v = v % R_MOD * (1 << 256) % R_MOD
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


_RINV = pow(1 << 256, -1, R_MOD)


def _fr_to_int(a):
    a = np.asarray(a, dtype=np.uint64).reshape(4)
    v = sum(int(a[i]) << (64 * i) for i in range(4))
    return v * _RINV % R_MOD


class _View:
    """a stretch of a pooled device allocation (same surface as api.DeviceBuffer; freeing it is a no-op)"""

    def __init__(self, buf, offset, nbytes):
        self.ptr, self.nbytes = ctypes.c_void_p(buf.ptr.value + int(offset)), int(nbytes)

    def at(self, offset):
        return ctypes.c_void_p(self.ptr.value + int(offset))

    def download(self, shape, dtype=np.uint64, offset=0):
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        check(api.init().vdb_memcpy_d2h(api._p(out), self.at(offset), _sz(out.nbytes)))
        return out

    def free(self):
        pass


class _Evals(Mapping):
    """evaluations as returned by prove(): (name, rotation) -> list of canonical Python integers, converted from the
    Montgomery arrays the device returned when first asked for (a proof of 2 x 10^4 columns carries ~2 x 10^5 of them)"""

    def __init__(self, raw):
        self.raw, self._ints = raw, {}

    def __getitem__(self, key):
        if key not in self._ints:
            a = np.ascontiguousarray(self.raw[key], dtype=np.uint64).reshape(-1, 4)
            self._ints[key] = [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in api.fr_to_canonical(a)] if len(a) else []
        return self._ints[key]

    def __iter__(self):
        return iter(self.raw)

    def __len__(self):
        return len(self.raw)


class _Poly:
    """a set of polynomials in the three forms the rounds use; any of them may be absent.  In a sharded proof the buffers hold
    this rank's part only: `ranges` = the [lo, hi) stretches of the set's global numbering that lie in the buffers one after the
    other (n_cols polynomials in all), `n_total` = the size of the whole set; `commits` is always the whole set's (gathered).
    `replicated`: every rank holds the whole set (the small fixed polynomials, the quotient's pieces)."""

    def __init__(self, name, n_cols, lag=None, coeff=None, ext=None, commits=None, ranges=None, n_total=None, replicated=False):
        self.name, self.n_cols, self.lag, self.coeff, self.ext, self.commits = name, n_cols, lag, coeff, ext, commits
        self.ranges = [(0, n_cols)] if ranges is None else [(int(a), int(b)) for a, b in ranges if b > a]
        self.n_total = n_cols if n_total is None else int(n_total)
        self.replicated = replicated
        assert sum(b - a for a, b in self.ranges) == n_cols

    def free(self):
        for b in (self.lag, self.coeff, self.ext):
            if b is not None:
                b.free()
        self.lag = self.coeff = self.ext = None


class ProverRounds:
    """`comm`: the exchange steps of a sharded proof (dist.Comm over torch.distributed: RCCL on a multi-GPU node, gloo in the
    tests) when the hot path `hp` holds one rank's column blocks (col_shard = (rank, world)); None for a proof on one GPU.
    Every rank runs the same rounds on its own columns and its own sets of the permutation argument (shardmap.ShardMap) and
    keeps a transcript of its own: commitments and evaluations are exchanged in the global order before they are absorbed, so
    every rank derives the same challenges and ends with the same proof bytes — the bytes a single GPU writes with the same
    blinding scalars."""

    def __init__(self, hp, block_cols=BLOCK_COLS, comm=None):
        from .dist import LocalComm
        from .shardmap import ShardMap
        self.degree = constraint_degree(hp.n_lk_cols)
        self.chunk_len, self.n_h = self.degree - 2, self.degree - 1
        self.n_slots = self.n_h
        assert block_cols % self.chunk_len == 0
        self.block_cols = block_cols
        self.hp, self.lib = hp, hp.lib
        self.comm = comm if comm is not None else LocalComm()
        self.rank, self.world = hp.rank, hp.world
        assert (self.comm.rank, self.comm.world) == (self.rank, self.world), "the hot path's shard and the communicator disagree"
        self.k, self.rows, self.ne = hp.k, hp.rows, hp.rows * self.n_slots      # ne: points per column of a coset array
        self.usable = hp.rows - N_BLIND
        self.n_adv, self.n_lk, self.n_cols = hp.n_adv_cols, hp.n_lk_cols, hp.n_cols
        self.n_perm = self.n_cols + 2                # the permutation argument's columns: advice, lookup, the constants' fixed column, the instance column
        self.n_sets = -(-self.n_perm // self.chunk_len)
        self.map = ShardMap(hp.shards, self.n_adv, self.n_lk, self.chunk_len)
        (self.a_lo, self.a_hi), (self.l_lo, self.l_hi) = hp.shards[self.rank]
        self.my_adv, self.my_lk = self.a_hi - self.a_lo, self.l_hi - self.l_lo
        self.set_ranges = self.map.set_ranges(self.rank)                       # the sets whose running products this rank computes
        self.sig_ranges = self.map.set_col_ranges(self.rank)                   # their columns: the sigma columns this rank keeps
        self.my_sets = sum(hi - lo for lo, hi in self.set_ranges)
        self.my_sig = sum(hi - lo for lo, hi in self.sig_ranges)
        self.adv_ranges = [(self.a_lo, self.a_hi), (self.n_adv + self.l_lo, self.n_adv + self.l_hi)]      # my columns among all advice + lookup columns
        self.foreign = self.map.foreign_cols(self.rank)                        # columns of my sets that another rank holds
        self.stray = self.map.stray_cols(self.rank)                            # my columns that lie in another rank's set
        self.delta = api.fr_delta()
        self.fixed = {}
        self._vk_digest = None
        self.d_map32 = None
        self.d_inst_cells = None
        self.public_cells = []          # the circuit's default public cells (circuit_map); a key loaded from a file brings its own
        self.instance_cells = []

    # ------------------------------------------------------------------ local positions of global things
    def _set_local(self, i):
        """position of global set i in this rank's product buffers"""
        off = 0
        for lo, hi in self.set_ranges:
            if lo <= i < hi:
                return off + i - lo
            off += hi - lo
        raise KeyError(i)

    def _sig_local(self, p):
        """position of permutation column p in this rank's sigma buffers"""
        off = 0
        for lo, hi in self.sig_ranges:
            if lo <= p < hi:
                return off + p - lo
            off += hi - lo
        raise KeyError(p)

    def _col_local(self, p):
        """position of advice / lookup column p (permutation numbering) in the hot path's column buffer [my advice | my lookup]; None: not mine"""
        if self.a_lo <= p < self.a_hi:
            return p - self.a_lo
        if self.l_lo <= p - self.n_adv < self.l_hi:
            return self.my_adv + p - self.n_adv - self.l_lo
        return None

    def _globalize(self, local, ranges, n_total):
        """rows this rank made, placed at `ranges` of an array of n_total rows; the other ranks' rows come with the exchange"""
        local = np.ascontiguousarray(local, dtype=np.uint64)
        if self.world == 1:
            return local
        out = np.zeros((n_total,) + local.shape[1:], dtype=np.uint64)
        off = 0
        for lo, hi in ranges:
            out[lo:hi] = local[off: off + hi - lo]
            off += hi - lo
        return self.comm.sum_disjoint(out)

    # ------------------------------------------------------------------ helpers on device-resident columns
    def _to_coeff(self, lag_buf, n_cols):
        """copy of Lagrange-form columns turned into coefficients"""
        c = api.DeviceBuffer(max(n_cols, 1) * self.rows * B)
        check(self.lib.vdb_memcpy_d2d(c.ptr, lag_buf.ptr, _sz(n_cols * self.rows * B)))
        check(self.lib.vdb_lagrange_to_coeff_dev(c.ptr, _sz(n_cols), self.k))
        return c

    def _to_ext(self, coeff_buf, n_cols):
        e = api.DeviceBuffer(max(n_cols, 1) * self.ne * B)
        check(self.lib.vdb_coeff_to_cosets_dev(coeff_buf.ptr, e.ptr, _sz(n_cols), self.k, self.n_slots, None))
        return e

    def _srs_for(self, n_cols, basis, dense):
        return self.hp.srs if basis == 1 and not dense else (self.srs_few if basis == 0 and n_cols <= 8 else self.srs_m)

    def _commit(self, buf, n_cols, basis, dense=True):
        out = np.zeros((n_cols, 8), dtype=np.uint64)
        if n_cols:
            check(self.lib.vdb_msm_batch_dev(self._srs_for(n_cols, basis, dense).h, basis, buf.ptr, _sz(n_cols), _sz(self.rows), api._p(out)))
        return out

    def _commit_begin(self, buf, n_cols, basis, dense=True):
        """queue the commitment of n_cols columns and return (vdb_msm_batch_masked_dev_begin); _commit_end collects the points"""
        if n_cols:
            check(self.lib.vdb_msm_batch_masked_dev_begin(self._srs_for(n_cols, basis, dense).h, basis, buf.ptr, _sz(n_cols), _sz(self.rows), None, None))

    def _commit_end(self, n_cols):
        out = np.zeros((n_cols, 8), dtype=np.uint64)
        if n_cols:
            check(self.lib.vdb_msm_batch_end(api._p(out), _sz(n_cols)))
        return out

    def _blind(self, buf, n_cols, from_row, seed, ranges=None, n_total=None):
        """uniform field elements into the rows from `from_row` on of every column (halo2 fills them with Scalar::random(rng)
        from OsRng): 64 bytes of entropy each, reduced on the device; `seed` = None draws from the operating system.
        `ranges` / `n_total`: the n_cols columns are those stretches of a set of n_total columns; a seeded stream (test hook) is
        then drawn for the whole set and this rank's part taken, so that a sharded proof blinds as the unsharded one does."""
        cnt = self.rows - from_row
        if n_cols * cnt == 0:
            return
        if seed is None or ranges is None or n_total == n_cols:
            d = api.DeviceBuffer(n_cols * cnt * B)
            api.random_scalars_dev(d.ptr, n_cols * cnt, seed=seed)
            check(self.lib.vdb_fill_rows_dev(buf.ptr, _sz(n_cols), _sz(self.rows), _sz(from_row), d.ptr))
        else:
            d = api.DeviceBuffer(n_total * cnt * B)
            api.random_scalars_dev(d.ptr, n_total * cnt, seed=seed)
            off = 0
            for lo, hi in ranges:
                if hi > lo:
                    check(self.lib.vdb_fill_rows_dev(buf.at(off * self.rows * B), _sz(hi - lo), _sz(self.rows), _sz(from_row), d.at(lo * cnt * B)))
                off += hi - lo
        api.sync()
        d.free()

    def vk_digest(self):
        """The verifying key's entry into the transcript: ONE scalar, as halo2 absorbs vk.transcript_repr (a hash of the key made
        at keygen) — here the squeeze of a sponge of its own over every fixed commitment in FIXED order, computed once per key (the
        38 k points of the k = 16 cosine circuit would otherwise cost every proof 19 k permutations on the host)."""
        if self._vk_digest is None:
            tr0 = api.Transcript()
            for name in FIXED:
                tr0.common_points(self.fixed[name].commits)
            self._vk_digest = tr0.squeeze()
            tr0.free()
        return self._vk_digest

    def _fixed_poly(self, name, lag_buf, n_cols, keep_lag=True, keep_ext=True, ranges=None, n_total=None):
        """commitment and coefficient form of a fixed polynomial; its extended coset only when it is small (the selector and
        sigma cosets — 4x the columns — are produced block by block inside the quotient instead of being held).  With `ranges`
        the buffer holds this rank's part of a set of n_total polynomials: every rank commits its part, the commitments of the
        whole set are exchanged (they make the verifying key's digest); without, every rank holds the (small) whole."""
        commits = self._commit(lag_buf, n_cols, 1)
        if ranges is not None:
            commits = self._globalize(commits, ranges, n_total)
        if keep_lag:
            coeff = self._to_coeff(lag_buf, n_cols)
        else:                      # transformed where it lies: no second buffer of tens of GB for the sigma columns and the selectors
            check(self.lib.vdb_lagrange_to_coeff_dev(lag_buf.ptr, _sz(n_cols), self.k))
            coeff, lag_buf = lag_buf, None
        p = _Poly(name, n_cols, lag=lag_buf, coeff=coeff, ext=self._to_ext(coeff, n_cols) if keep_ext else None, commits=commits,
                  ranges=ranges, n_total=n_total, replicated=ranges is None)
        self.fixed[name] = p
        return p

    # ------------------------------------------------------------------ keygen side (untimed): the fixed polynomials
    def circuit_map(self, d_flags):
        """The circuit's constraint map (circuit_sym.CopyMap) for the gadget this hot path runs: the symbolic trace of the
        fixed-point gadgets, or of the Poseidon sponge for the Merkle circuit."""
        from .pipeline import DistancesHotPath, FixedPointHotPath, MerkleHotPath, NearestHotPath, QueryHotPath
        hp = self.hp
        on_device = getattr(self, "map_on_device", True)
        if isinstance(hp, FixedPointHotPath):
            # one value, a handful of FixedPointChip calls (examples/fixed_point.rs): x and every result public
            from . import circuit_sym as CS
            cm, outs = CS.trace_fixed_point(hp.ops, hp.P, hp.L)
            self.public_cells = [int(c) for c in outs]
            return cm
        if isinstance(hp, DistancesHotPath):
            # two vectors, a handful of distances: the whole trace on the host (examples/distances.rs, examples/euclid.rs)
            from . import circuit_sym as CS
            cm, outs = CS.trace_distances(hp.metrics, hp.dim, hp.P, hp.L)
            self.public_cells = [int(c) for c in outs] if hp.public else []      # examples/distances.rs:44-59 make_public.push(dist)
            return cm

        def fetch(lo, hi):
            c = api.fr_to_canonical(hp.d_stream.download((hi - lo, 4), offset=lo * B))
            return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in c]

        def fetch_flags(lo, hi):
            return d_flags.download((hi - lo,), dtype=np.uint8, offset=lo)
        if isinstance(hp, MerkleHotPath):
            if on_device:
                from .circuit_dev import DeviceBuilder, place_merkle
                bld = DeviceBuilder(hp.n_cells, 0)
                self.root_cell, end = place_merkle(bld, hp.n, hp.dim, hp.n_in, 0, fetch_flags, fetch)
                assert end == hp.n_cells
                cm = bld.finish()
            else:
                from .copymap import merkle_circuit_map
                cm, self.root_cell = merkle_circuit_map(hp.n, hp.dim, d_flags.download((hp.n_cells,), dtype=np.uint8), fetch)
            self.public_cells = [int(self.root_cell)]                               # examples/merkle.rs:47 make_public.push(root)
            return cm
        from . import circuit_sym as CS
        from .circuit_dev import DeviceBuilder
        # the unit blocks are traced on the host (a few thousand cells each); their hundreds of thousands of instances are placed
        # by the device (circuit_dev.py): the map's 10^9-cell arrays never exist on the host
        builder = DeviceBuilder if on_device else None
        if isinstance(hp, QueryHotPath):
            # nearest_vector, then merkle_commitment over the same assigned vectors, in one map (examples/query.rs)
            from .circuit_dev import place_merkle
            bld, (_ind, res), used = CS.build_nearest(hp.metric_name, hp.n, hp.dim, hp.P, hp.L, builder=DeviceBuilder, extra_cells=hp.merkle_cells, finish=False)
            assert used == hp.n_in + hp.nearest_cells
            self.root_cell, end = place_merkle(bld, hp.n, hp.dim, used, hp.dim, fetch_flags, fetch)
            assert end == hp.n_cells
            cm = bld.finish()
            self.public_cells = [int(c) for c in res] + [int(self.root_cell)]       # examples/query.rs:58, :69: the result vector, then the root
            return cm
        if isinstance(hp, NearestHotPath):
            cm, (_ind, res) = CS.build_nearest(hp.metric_name, hp.n, hp.dim, hp.P, hp.L, builder=builder)
            self.public_cells = [int(c) for c in res]                               # examples/query.rs:58 make_public.extend(result)
        else:
            cm, (cent, _ind) = CS.build_kmeans(hp.metric_name, hp.n, hp.dim, hp.K, hp.I, hp.P, hp.L, builder=builder)
            self.public_cells = [int(c) for c in np.asarray(cent).reshape(-1)]      # examples/kmeans.rs:51-56: every centroid, word by word
        return cm

    def keygen(self, circuit=None, instance_cells=None, check=True):
        """The Keygen arm's work for the rounds (src/scaffold/mod.rs:267-283 -> keygen_vk / keygen_pk): gate selectors, the
        permutation (sigma columns) from the circuit's constraint map, the constants' fixed column, the range table.
        `circuit`: a circuit_sym.CopyMap over the stream cells; None = the map of the gadget this hot path runs (circuit_map).
        `instance_cells`: the stream cells the closure makes public, in `make_public` order; cell i is tied to row i of the
        instance column by a copy constraint (src/scaffold/mod.rs:400).  None = what the reference's example of this gadget
        exposes (circuit_map: centroids / result vector / root) when the map is the gadget's own, nothing for a map handed in.
        `check`: run the device-side MockProver on the keygen witness with the whole map (vdb_mock_check_dev); the report is
        kept in self.keygen_report (a circuit the witness does not satisfy can still be set up — the proof will not verify)."""
        hp, lib, rows, k = self.hp, self.lib, self.rows, self.k
        # the fixed columns' commitments work in the bounded MSM work space of setup (pipeline.setup); the first step() lifts the bound
        api.msm_scratch_cap(api.KEYGEN_SCRATCH_CAP)
        hp._cap_lifted = False
        # the derived columns (products, quotient, opening quotients) and the fixed sigma columns hold full-width scalars
        self.srs_m = api.Srs(k, hp.g_monomial, hp.g_lagrange, window_bits=14)
        self.srs_few = api.Srs(k, hp.g_monomial, None)     # a handful of columns: the bucket folding dominates, fewer buckets win
        # gate selectors from a flag-recording witness run
        d_flags = hp.keygen_flags()
        from ._lib import check as _chk     # (`check` is this method's flag)
        self.public_cells = []
        cm = circuit if circuit is not None else self.circuit_map(d_flags)
        if instance_cells is None:
            instance_cells = self.public_cells
        self.instance_cells = [int(c) for c in instance_cells]
        if len(self.instance_cells) > self.usable or any(not 0 <= c < hp.n_cells for c in self.instance_cells):
            raise ValueError("public cells outside the stream, or more of them than usable rows of the instance column")
        on_dev = hasattr(cm, "d_copy_of")
        n_src = cm.n_lookup if on_dev else (None if cm.lookup_src is None else len(cm.lookup_src))
        if cm.n_cells != hp.n_cells or (n_src is not None and n_src != hp.n_lookup):
            raise ValueError("the constraint map does not describe this circuit (cell counts differ)")
        self.circuit = cm
        self.consts = [int(v) for v in cm.consts]
        if len(self.consts) > self.usable:
            raise ValueError("more distinct constants than usable rows of the fixed column")
        if check:
            self.keygen_report = self.mock_check(d_flags)
        # sigma columns over [advice | lookup | constants | instance]: the cycles of the copy classes, built on the device
        # (vdb_permutation_mapping_dev: pointer jumping, one radix sort; copymap.mapping_from_copy_of is the host restatement
        # the tests compare it with).  A map without lookup sources leaves the lookup columns untied: only the tests' negative
        # cases want that.
        from .pipeline import MINIMUM_ROWS
        d_parent = api.DeviceBuffer(hp.n_cells * 8)
        d_lsrc = None
        if on_dev:
            bad, nosrc = ctypes.c_uint64(), ctypes.c_uint64()
            _chk(lib.vdb_copymap_finish_dev(cm.d_copy_of.ptr, cm.d_const_idx.ptr, ctypes.c_uint64(hp.n_cells), cm.d_lookup_src.ptr, ctypes.c_uint64(hp.n_lookup),
                                            d_parent.ptr, ctypes.byref(bad), ctypes.byref(nosrc)))
            if bad.value:
                raise ValueError("a cell tied to a constant must be the root of its copies")
            tie_lookups = bool(hp.n_lookup)
            lsrc_ptr = cm.d_lookup_src.ptr
        else:
            parent = cm.copy_of.astype(np.int64, copy=True)
            tied = cm.const_idx >= 0
            if (parent[tied] != np.flatnonzero(tied)).any():
                raise ValueError("a cell tied to a constant must be the root of its copies")
            parent[tied] = hp.n_cells + cm.const_idx[tied]
            d_parent.upload(parent)
            del parent
            tie_lookups = bool(hp.n_lookup) and cm.lookup_src is not None
            if tie_lookups:
                d_lsrc = api.DeviceBuffer(hp.n_lookup * 8)
                d_lsrc.upload(np.ascontiguousarray(cm.lookup_src, dtype=np.int64))
            lsrc_ptr = d_lsrc.ptr if tie_lookups else None
        d_map = api.DeviceBuffer(self.n_perm * rows * 8)
        bp64 = np.ascontiguousarray(hp.bp, dtype=np.uint64)
        self._upload_instance_cells()
        # The key's two big buffers — the sigma columns of my sets, the selectors of my advice columns — are ONE allocation, made
        # before the permutation is built and lent to it for its sort records (51 GiB at C4'): HBM that is mapped once and never handed
        # back (mapping or clearing HBM costs this driver ~30 ms / GiB, vdb_alloc_stats).
        if getattr(self, "d_key", None) is not None:
            self.d_key.free()
        self.d_key = api.DeviceBuffer(max(self.my_sig + self.my_adv, 1) * rows * B)
        _chk(lib.vdb_permutation_mapping_ws_dev(d_parent.ptr, ctypes.c_uint64(hp.n_cells), ctypes.c_uint64(len(self.consts)), api._p(bp64), ctypes.c_uint64(len(bp64)), k,
                                                lsrc_ptr if tie_lookups else None, ctypes.c_uint64(hp.n_lookup if tie_lookups else 0),
                                                ctypes.c_uint64(rows - MINIMUM_ROWS), ctypes.c_uint64(self.n_cols),
                                                self.d_inst_cells.ptr, ctypes.c_uint64(len(self.instance_cells)), d_map.ptr,
                                                self.d_key.ptr, _sz(self.d_key.nbytes)))
        d_parent.free()
        if d_lsrc is not None:
            d_lsrc.free()
        if on_dev and hp.n_cells >= (1 << 28) and not getattr(self, "keep_circuit", False):
            cm.free()                    # tens of GB at BASELINE sizes: the proof needs the room; small circuits keep their map (tests, mock_check)
        self._d_map_for_tests = d_map if getattr(self, "keep_mapping", False) else None
        # the mapping stays with the key in 32 bits per cell when column and row fit: the product round makes the sigma columns'
        # Lagrange form from it (one product per cell) instead of transforming their coefficient form back
        self.d_map32 = None
        if (self.n_perm - 1).bit_length() + k <= 32 and (getattr(self, "keep_packed_mapping", True) or self.world > 1):
            self.d_map32 = api.DeviceBuffer(self.n_perm * rows * 4)
            _chk(lib.vdb_permutation_mapping_pack_dev(d_map.ptr, _sz(self.n_perm), k, self.d_map32.ptr))
            api.sync()
        if self.world > 1 and self.d_map32 is None:
            raise ValueError("a sharded key keeps the packed mapping: column and row of a cell must fit 32 bits")
        # sigma columns: every rank builds the cycles of the whole circuit (the copy classes cross all columns) and keeps the
        # columns of its own sets
        d_sigma = _View(self.d_key, 0, self.my_sig * rows * B)
        off = 0
        for lo, hi in self.sig_ranges:
            _chk(lib.vdb_permutation_sigma_dev(d_map.at(lo * rows * 8), _sz(hi - lo), k, api._p(self.delta), d_sigma.at(off * rows * B))
                 if self.world == 1 else
                 lib.vdb_permutation_sigma_packed_dev(self.d_map32.at(lo * rows * 4), _sz(hi - lo), _sz(self.n_perm), k, api._p(self.delta), d_sigma.at(off * rows * B)))
            off += hi - lo
        if self._d_map_for_tests is None:
            d_map.free()
        self._fixed_poly("sigma", d_sigma, self.my_sig, keep_lag=False, keep_ext=False, ranges=self.sig_ranges, n_total=self.n_perm)
        # gate selectors (after the permutation's work space is gone: both are tens of GB at BASELINE sizes), of my advice columns
        d_mine = _View(self.d_key, self.my_sig * rows * B, self.my_adv * rows * B)
        if self.my_adv == self.n_adv:
            d_q = d_mine
            _chk(lib.vdb_layout_selectors_dev(d_flags.ptr, ctypes.c_uint64(hp.n_cells), api._p(hp.bp), ctypes.c_uint64(len(hp.bp)), k, d_q.ptr))
            d_flags.free()
        else:       # a rank of a sharded run lays out every column's selectors and keeps its own
            d_q = api.DeviceBuffer(self.n_adv * rows * B)
            _chk(lib.vdb_layout_selectors_dev(d_flags.ptr, ctypes.c_uint64(hp.n_cells), api._p(hp.bp), ctypes.c_uint64(len(hp.bp)), k, d_q.ptr))
            d_flags.free()
            _chk(lib.vdb_memcpy_d2d(d_mine.ptr, d_q.at(self.a_lo * rows * B), _sz(self.my_adv * rows * B)))
            api.sync()
            d_q.free()
            d_q = d_mine
        self._fixed_poly("sel", d_q, self.my_adv, keep_lag=False, keep_ext=False, ranges=[(self.a_lo, self.a_hi)], n_total=self.n_adv)
        # the constants' fixed column: constant r at row r (halo2-base assigns the distinct constants of a circuit to fixed cells
        # and ties every Constant advice cell to its fixed cell through the permutation)
        cst = np.zeros((rows, 4), dtype=np.uint64)
        if self.consts:
            cst[: len(self.consts)] = api.fr_from_canonical(np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in self.consts], dtype=np.uint64))
        d_cst = api.DeviceBuffer(rows * B)
        d_cst.upload(cst)
        self._fixed_poly("cst", d_cst, 1)
        # its coset sits behind the advice cosets, the instance column's behind it (the permutation term of the quotient reads one
        # contiguous block of columns)
        if hp.ext_cols >= self.my_adv + self.my_lk + 2:
            _chk(lib.vdb_memcpy_d2d(hp.d_ext.at((self.my_adv + self.my_lk) * self.ne * B), self.fixed["cst"].ext.ptr, _sz(self.ne * B)))
        # range table 0 .. 2^L - 1, zero below; Lagrange selectors l0, l_last, l_active
        tab = np.arange(rows, dtype=np.uint64)
        tab[tab >= (1 << hp.L)] = 0
        d_tab = api.DeviceBuffer(rows * B)
        d_tab.upload(api.fr_from_canonical(np.stack([tab, np.zeros_like(tab), np.zeros_like(tab), np.zeros_like(tab)], axis=1)))
        self._fixed_poly("table", d_tab, 1)
        lag = np.zeros((3, rows, 4), dtype=np.uint64)
        one = _fr_from_int(1)
        lag[0, 0], lag[1, self.usable], lag[2, : self.usable] = one, one, one
        d_l = api.DeviceBuffer(lag.nbytes)
        d_l.upload(lag)
        self._fixed_poly("lag", d_l, 3)
        api.sync()
        return self._alloc_working_set()

    def _upload_instance_cells(self):
        cells = np.asarray(self.instance_cells, dtype=np.int64)
        if getattr(self, "d_inst_cells", None) is not None:      # a second keygen / key load on the same object
            self.d_inst_cells.free()
        self.d_inst_cells = api.DeviceBuffer(max(cells.nbytes, 32))
        if cells.nbytes:
            self.d_inst_cells.upload(cells)

    def mock_check(self, d_flags=None, instances=None):
        """The Mock stage on the witness in HBM with this circuit's whole constraint map (vdb_mock_check_dev): gate rows, the
        range table, every copy, every lookup source, every constant and asserted constant — and, with `instances` (canonical
        integers, one per public cell: MockProver::run's third argument), every public cell against the value claimed for it.
        api.MockReport."""
        hp = self.hp
        cm = self.circuit
        own = d_flags is None
        if own:
            d_flags = hp.keygen_flags()
            shard, hp.shard_witness = hp.shard_witness, False      # the check walks the whole witness, whatever this rank's columns
            try:
                hp._witness()
            finally:
                hp.shard_witness = shard
        bufs = []

        def dev(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            b = api.DeviceBuffer(max(a.nbytes, 32))
            if a.nbytes:
                b.upload(a)
            bufs.append(b)
            return b
        try:
            if hasattr(cm, "d_copy_of"):
                if cm.d_copy_of is None:
                    raise RuntimeError("the circuit's constraint map was released after keygen (set keep_circuit = True before keygen to keep it)")
                d_copy, d_cidx, d_lsrc = cm.d_copy_of, cm.d_const_idx, cm.d_lookup_src if hp.n_lookup else None
            else:
                d_copy = dev(cm.copy_of, np.int64)
                d_lsrc = dev(cm.lookup_src, np.int64) if hp.n_lookup and cm.lookup_src is not None else None
                d_cidx = dev(cm.const_idx, np.int64)
            tab = np.zeros((max(len(cm.consts), 1), 4), dtype=np.uint64)
            if len(cm.consts):
                tab[: len(cm.consts)] = api.fr_from_canonical(np.array([[(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in cm.consts], dtype=np.uint64))
            d_tab = dev(tab, np.uint64)
            rep = api.mock_check_dev(hp.d_stream.ptr, hp.n_cells, d_flags.ptr, hp.d_lookup.ptr, hp.n_lookup, hp.L, d_copy.ptr,
                                     None if d_lsrc is None else d_lsrc.ptr, None, d_cidx.ptr, d_tab.ptr, len(cm.consts))
            if instances is not None:
                if len(instances) != len(self.instance_cells):
                    raise ValueError("one value per public cell")
                vals = np.zeros((max(len(instances), 1), 4), dtype=np.uint64)
                if len(instances):
                    vals[: len(instances)] = np.stack([_fr_from_int(int(v)) for v in instances])
                api.mock_check_instances_dev(rep, hp.d_stream.ptr, hp.n_cells, dev(self.instance_cells, np.int64).ptr, dev(vals, np.uint64).ptr, len(instances))
            return rep
        finally:
            for b in bufs:
                b.free()
            if own:
                d_flags.free()

    def mock_check_instances(self, instances):
        """The `instances` argument of MockProver::run alone, on the witness in HBM (vdb_mock_check_instances_dev): public cell i of
        keygen must hold instances[i] (canonical integers).  Needs no constraint map, so it also serves after keygen released the
        map of a BASELINE-size circuit.  api.MockReport with only the instance fields filled."""
        if len(instances) != len(self.instance_cells):
            raise ValueError("one value per public cell")
        rep = api.MockReport()
        if not len(instances):
            return rep
        vals = np.stack([_fr_from_int(int(v)) for v in instances])
        d_vals = api.DeviceBuffer(vals.nbytes)
        try:
            d_vals.upload(vals)
            api.mock_check_instances_dev(rep, self.hp.d_stream.ptr, self.hp.n_cells, self.d_inst_cells.ptr, d_vals.ptr, len(instances))
        finally:
            d_vals.free()
        return rep

    def _alloc_working_set(self):
        lib, rows, CHUNK_LEN = self.lib, self.rows, self.chunk_len
        # The working set of prove(), allocated once (device allocations of tens of GB take seconds): the derived columns
        # [pa | ps | zp | zl] (Lagrange, then coefficient form in place), the lookup columns laid out, and block buffers of
        # `block_cols` columns — two in Lagrange form (the permutation's columns and their sigma columns), two of extended
        # cosets, one of product cosets — through which every per-column stage streams.  The library's MSM scratch is released
        # first so that it is re-sized to what is left.
        if api.scratch_held() > 2 * api.KEYGEN_SCRATCH_CAP:      # a hot path that has proved already holds the prover's big work space
            check(lib.vdb_scratch_release())
        n_der = 3 * self.my_lk + self.my_sets
        blk = max(2 * CHUNK_LEN, min(self.block_cols, -(-max(self.my_sig, 1) // (2 * CHUNK_LEN)) * (2 * CHUNK_LEN)) // (2 * CHUNK_LEN) * (2 * CHUNK_LEN))
        self.blk_alloc = blk
        self.pool_der = api.DeviceBuffer(max(n_der, 1) * rows * B)
        self.d_lklag = api.DeviceBuffer(max(self.my_lk, 1) * rows * B)
        # what a sharded proof receives from other ranks: the columns that complete the set spanning the advice / lookup junction
        # (Lagrange and coefficient form), one boundary product polynomial per requested set
        self.z_req = self.map.z_requests(self.rank)
        self.d_foreign_lag = api.DeviceBuffer(max(len(self.foreign), 1) * rows * B)
        self.d_foreign_coeff = api.DeviceBuffer(max(len(self.foreign), 1) * rows * B)
        self.d_zhalo = api.DeviceBuffer(max(len(self.z_req), 1) * rows * B)
        self.d_lag_a, self.d_lag_s = api.DeviceBuffer(blk * rows * B), api.DeviceBuffer(blk * rows * B)
        self.d_ea, self.d_eb = api.DeviceBuffer(blk * self.ne * B), api.DeviceBuffer(blk * self.ne * B)
        self.d_ez = api.DeviceBuffer((blk // CHUNK_LEN + 1) * self.ne * B)
        self.d_zf, self.d_zlast = api.DeviceBuffer(self.ne * B), api.DeviceBuffer(self.ne * B)
        self.d_h, self.d_h2, self.d_h3, self.d_h4 = (api.DeviceBuffer(self.ne * B) for _ in range(4))
        self.d_hg = api.DeviceBuffer(GATE_SLOTS * rows * B)         # the gates' accumulator, on the coset of 2 n points
        self.d_comb, self.d_quot = api.DeviceBuffer(rows * B), api.DeviceBuffer(rows * B)
        # the instance column: Lagrange form (the public values in rows 0 .. n_instances - 1, zero below), coefficients, coset —
        # made anew for every proof from the values of the statement
        self.d_inst_lag, self.d_inst_coeff, self.d_inst_ext = api.DeviceBuffer(rows * B), api.DeviceBuffer(rows * B), api.DeviceBuffer(self.ne * B)
        self.d_rand, self.d_hf = api.DeviceBuffer(rows * B), api.DeviceBuffer(rows * B)       # the vanishing argument's random polynomial, h folded at x
        check(lib.vdb_memset_dev(self.d_inst_lag.ptr, 0, _sz(rows * B)))
        self._vk_digest = None
        self.vk_digest()            # part of the key, made here so that no proof pays for it
        # The cosets of the fixed polynomials that every proof would otherwise transform again — the sigma columns (n_slots cosets
        # each) and the selectors (two) — stay in HBM when they fit beside everything above with room to spare for the MSM's work space:
        # halo2's ProvingKey holds them too (fixed_cosets, permutation cosets).  They fit on a rank of a multi-GPU job and for circuits
        # up to a few thousand columns; BASELINE C4' on one card (20,969 columns: 126 + 36 GB) streams them block by block as before.
        self.fixed_cosets_resident = False
        need = (self.my_sig * self.ne + self.my_adv * GATE_SLOTS * rows) * B
        free, _total = api.mem_info()
        free += api.scratch_held()          # the work space already allocated counts towards the reserve kept for it
        reserve = int(os.environ.get("VDB_FIXED_COSETS_RESERVE_GB", "48")) << 30
        if os.environ.get("VDB_FIXED_COSETS", "1") != "0" and free - need >= reserve:
            sig, sel = self.fixed["sigma"], self.fixed["sel"]
            sig.ext = api.DeviceBuffer(max(self.my_sig * self.ne, 1) * B)
            check(lib.vdb_coeff_to_cosets_dev(sig.coeff.ptr, sig.ext.ptr, _sz(self.my_sig), self.k, self.n_slots, None))
            sel.ext = api.DeviceBuffer(max(self.my_adv * GATE_SLOTS * rows, 1) * B)
            check(lib.vdb_coeff_to_cosets_dev(sel.coeff.ptr, sel.ext.ptr, _sz(self.my_adv), self.k, GATE_SLOTS, None))
            api.sync()
            self.fixed_cosets_resident = True
        return self

    # ------------------------------------------------------------------ the proving key on disk
    def save_verifying_key(self, path, opened=None):
        """What the Keygen arm writes beside the proving key (src/scaffold/mod.rs:276-281: data/{name}.vk) in this build's own
        container (upstream's is SerdeFormat::RawBytes of halo2's VerifyingKey: parity unpinned): the circuit's shape, the commitments
        of the fixed polynomials in FIXED order, the transcript digest made of them, the SRS scalar of the deterministic "unsafe"
        setup the reference's gen_srs uses (a verifier derives [tau]_2 from it; a ceremony SRS would carry the G2 point instead), and
        — when a proof's `opened` map is given — which polynomial is opened at which rotation.  io.read_verifying_key reads it."""
        from .io import write_verifying_key
        meta = dict(rows=self.rows, k=self.k, n_adv=self.n_adv, n_lk=self.n_lk, n_cols=self.n_cols, n_sets=self.n_sets, chunk_len=self.chunk_len,
                    n_blind=N_BLIND, delta=str(_fr_to_int(self.delta)), n_instances=len(self.instance_cells), tau=str(self.hp.tau),
                    vk_digest=str(_fr_to_int(self.vk_digest())))
        if opened is not None:
            meta["opened"] = {str(rot): list(names) for rot, names in opened.items()}
        write_verifying_key(path, meta, {name: self.fixed[name].commits for name in FIXED})

    def _key_header(self):
        """what a proving-key file must agree on with the run that loads it: the circuit's shape and, for a sharded key, this rank's
        place in the partition (rank, world, every rank's column blocks)"""
        shards = np.asarray([[a[0], a[1], l[0], l[1]] for a, l in self.hp.shards], dtype=np.int64)
        return (np.array([self.k, self.n_adv, self.n_lk, self.hp.L, self.chunk_len, N_BLIND, self.rank, self.world], dtype=np.uint64), shards)

    def save_proving_key(self, path):
        """What the reference's Keygen arm leaves for the Prove arm (src/scaffold/mod.rs:272-281: pinning + pk), in this
        build's own container: an .npz (numpy.load with allow_pickle=False reads it) holding the circuit's shape, the break
        points, and per fixed polynomial — gate selectors, sigma columns, constants, range table, Lagrange selectors — its coefficient
        form and its commitments.  Upstream's pk file format (SerdeFormat::RawBytes) is not reproduced: parity unpinned.
        Sharded (world > 1): every rank writes a file of its own (the caller names it per rank) with the selectors of its advice
        columns and the sigma columns of its sets, beside the whole set's commitments and the replicated small polynomials; the
        header records the partition, and load_proving_key refuses a file written for another rank or other column blocks."""
        api.sync()
        meta, shards = self._key_header()
        doc = {"meta": meta, "shards": shards, "break_points": np.asarray(self.hp.bp, dtype=np.uint64)}
        doc["instance_cells"] = np.asarray(self.instance_cells, dtype=np.int64)
        doc["instance_cells_are_the_default"] = np.array([int(self.instance_cells == [int(c) for c in self.public_cells])], dtype=np.int64)
        for name, q in self.fixed.items():
            doc[name + "_coeff"] = q.coeff.download((max(q.n_cols, 1), self.rows, 4))[: q.n_cols]
            doc[name + "_commits"] = q.commits
        np.savez(path, **doc)

    def load_proving_key(self, path):
        """The Prove arm's side: the fixed polynomials from a file written by save_proving_key instead of a keygen run (no
        flag-recording witness pass, no permutation construction).  The Lagrange forms the rounds read (sigma, table) are
        recovered with a forward transform; raises ValueError when the file describes another circuit, another rank or another
        partition of the columns."""
        with np.load(path, allow_pickle=False) as doc:
            meta, shards = self._key_header()
            if not np.array_equal(doc["meta"], meta) or not np.array_equal(doc["shards"], shards) \
                    or not np.array_equal(doc["break_points"], np.asarray(self.hp.bp, dtype=np.uint64)):
                raise ValueError("proving key does not describe this circuit (shape, break points, rank or column blocks differ)")
            return self._install_key(doc)

    def _install_key(self, doc):
        """doc[name + "_coeff"], doc[name + "_commits"] for every fixed polynomial, doc["instance_cells"]: the key's polynomials go
        to the device in the forms the rounds read"""
        hp, lib, rows, k = self.hp, self.lib, self.rows, self.k
        self.srs_m = api.Srs(k, hp.g_monomial, hp.g_lagrange, window_bits=14)
        self.srs_few = api.Srs(k, hp.g_monomial, None)
        omega = api.root_of_unity(k)
        # (name, polynomials held here, their place in the whole set, size of the whole set, Lagrange form kept, cosets kept)
        plan = (("sel", self.my_adv, [(self.a_lo, self.a_hi)], self.n_adv, False, False), ("sigma", self.my_sig, self.sig_ranges, self.n_perm, False, False),
                ("cst", 1, None, 1, True, True), ("table", 1, None, 1, True, True), ("lag", 3, None, 3, False, True))
        for name, n_cols, ranges, n_total, need_lag, keep_ext in plan:
            coeff_h = np.ascontiguousarray(doc[name + "_coeff"])
            commits = np.ascontiguousarray(doc[name + "_commits"])
            if coeff_h.shape != (n_cols, rows, 4) or commits.shape != (n_total, 8):
                raise ValueError("proving key: wrong shape for " + name)
            coeff = api.DeviceBuffer(max(coeff_h.nbytes, 32))
            if coeff_h.nbytes:
                coeff.upload(coeff_h)
            lag = None
            if need_lag:
                lag = api.DeviceBuffer(coeff_h.nbytes)
                check(lib.vdb_memcpy_d2d(lag.ptr, coeff.ptr, _sz(coeff_h.nbytes)))
                check(lib.vdb_ntt_batch_dev(lag.ptr, _sz(n_cols), k, api._p(omega), 0))
            self.fixed[name] = _Poly(name, n_cols, lag=lag, coeff=coeff, ext=self._to_ext(coeff, n_cols) if keep_ext else None, commits=commits,
                                     ranges=ranges, n_total=n_total, replicated=ranges is None)
        self.instance_cells = [int(c) for c in doc["instance_cells"]]      # the public cells come with the key
        if int(doc["instance_cells_are_the_default"][0]):
            self.public_cells = list(self.instance_cells)
        self._upload_instance_cells()
        if hp.ext_cols >= self.my_adv + self.my_lk + 2:
            check(lib.vdb_memcpy_d2d(hp.d_ext.at((self.my_adv + self.my_lk) * self.ne * B), self.fixed["cst"].ext.ptr, _sz(self.ne * B)))
        api.sync()
        return self._alloc_working_set()

    # upstream's own key files ------------------------------------------------------------------------------------------------
    RAW_FIXED = ("table", "cst", "sel")     # halo2-base creates the fixed columns in this order (io.py, [UPSTREAM-RECALL])
    RAW_BLOCK = 64                          # polynomials converted and written at a time

    @property
    def raw_ext_k(self):
        """halo2's extended domain has 2^(k + ceil(log2(degree - 1))) points: 4n for the circuits with lookups (degree 4), 2n without
        (the rounds here evaluate on degree - 1 cosets of size n instead, self.ne points per column)"""
        return int(self.chunk_len).bit_length()

    def _raw_blocks(self, names, form):
        """The polynomials of `names`, in order, as host blocks of at most RAW_BLOCK: form "coeff", "lagrange" (values over the
        2^k domain) or "extended" (values over the 4n coset, halo2's ExtendedLagrangeCoeff)"""
        omega = api.root_of_unity(self.k)
        for name in names:
            q = self.fixed[name]
            for lo in range(0, q.n_cols, self.RAW_BLOCK):
                m = min(self.RAW_BLOCK, q.n_cols - lo)
                coeff = q.coeff.download((m, self.rows, 4), offset=lo * self.rows * B)
                yield coeff if form == "coeff" else api.ntt_batch(coeff, omega) if form == "lagrange" else api.coeff_to_extended(coeff, self.raw_ext_k)

    def _write_vk_raw(self, f):
        from .io import write_vk_raw
        if self.world != 1:
            raise ValueError("upstream's key files describe the whole circuit: write them from a one-rank keygen (save_proving_key writes a rank's share)")
        api.sync()
        fixed = np.concatenate([self.fixed[name].commits for name in self.RAW_FIXED])
        write_vk_raw(f, self.k, fixed, self.fixed["sigma"].commits, (np.any(v != 0, axis=2) for v in self._raw_blocks(("sel",), "lagrange")))

    def save_verifying_key_raw(self, path):
        """data/{name}.vk as the reference's Keygen arm writes it (src/scaffold/mod.rs:276-281): halo2's
        `VerifyingKey::write(.., SerdeFormat::RawBytes)` layout (io.py restates it; [UPSTREAM-RECALL], parity unpinned) — k, the fixed
        columns' commitments (table, constants, one per gate selector), the sigma commitments, the selectors' bits.
        io.read_verifying_key_raw reads it back; tests/verify_file.py verifies a proof against it."""
        with open(path, "wb") as f:
            self._write_vk_raw(f)

    def save_proving_key_raw(self, path):
        """data/{name}.pk as snark-verifier-sdk's gen_pk leaves it for read_pk (src/scaffold/mod.rs:273, :325-331): halo2's
        `ProvingKey::write` layout — the verifying key; l_0, l_last, l_active_row over the extended domain; every fixed column as
        values, coefficients and extended coset; the sigma columns likewise.  Written a block of polynomials at a time (the cosets are
        made for the file only: 6 x 2^k x 32 bytes per column, which is why upstream's later versions dropped them from the key).
        The break points and the public cells are not part of it — they travel in the pinning file (io.write_pinning) and with the
        circuit, as upstream."""
        from .io import write_poly_raw, write_polys_raw
        with open(path, "wb") as f:
            self._write_vk_raw(f)
            for poly in next(self._raw_blocks(("lag",), "extended")):
                write_poly_raw(f, poly)
            n_fixed = sum(self.fixed[name].n_cols for name in self.RAW_FIXED)
            for form in ("lagrange", "coeff", "extended"):
                write_polys_raw(f, self._raw_blocks(self.RAW_FIXED, form), n_fixed)
            for form in ("lagrange", "coeff", "extended"):
                write_polys_raw(f, self._raw_blocks(("sigma",), form), self.n_perm)

    def load_proving_key_raw(self, path, instance_cells=None):
        """The Prove arm reading upstream's pk layout (custom_read_pk, src/scaffold/mod.rs:325-331): commitments and coefficient
        forms come from the file, the extended cosets are skipped over and re-derived on the device, l_0 / l_last / l_active are
        checked against this circuit's.  `instance_cells`: as in keygen (None = the gadget's own public cells, which a circuit map
        the circuit's constraint map names).  Raises ValueError when the file is not a key of this circuit."""
        from .io import read_vk_raw, read_poly_raw, read_polys_raw
        if self.world != 1:
            raise ValueError("upstream's key files describe the whole circuit: load them on one rank")
        rows, ne = self.rows, self.rows << self.raw_ext_k
        with open(path, "rb") as f:
            vk = read_vk_raw(f, self.n_perm, self.n_adv)
            if vk["k"] != self.k or len(vk["fixed_commitments"]) != self.n_adv + 2:
                raise ValueError("proving key does not describe this circuit (k or the number of fixed columns differ)")
            lag_ext = np.stack([read_poly_raw(f, ne) for _ in range(3)])
            read_polys_raw(f, self.n_adv + 2, rows, keep=False)
            fixed = read_polys_raw(f, self.n_adv + 2, rows)
            read_polys_raw(f, self.n_adv + 2, ne, keep=False)
            read_polys_raw(f, self.n_perm, rows, keep=False)
            sigma = read_polys_raw(f, self.n_perm, rows)
            read_polys_raw(f, self.n_perm, ne, keep=False)
            if f.read(1):
                raise ValueError("proving key: bytes after the permutation's cosets")
        lag = np.zeros((3, rows, 4), dtype=np.uint64)
        one = _fr_from_int(1)
        lag[0, 0], lag[1, self.usable], lag[2, : self.usable] = one, one, one
        lag_coeff = api.lagrange_to_coeff(lag)
        if not np.array_equal(api.coeff_to_extended(lag_coeff, self.raw_ext_k), lag_ext):
            raise ValueError("proving key: l_0, l_last, l_active_row are not this circuit's (another number of blinding rows?)")
        sel_bits = np.any(api.ntt_batch(fixed[2:], api.root_of_unity(self.k)) != 0, axis=2) if self.n_adv else np.zeros((0, rows), dtype=bool)
        if not np.array_equal(sel_bits, vk["selectors"]):
            raise ValueError("proving key: the selectors' bits are not where the selector columns are non-zero")
        if instance_cells is None:
            if not self.public_cells:
                d_flags = self.hp.keygen_flags()       # the circuit's own public cells: from its constraint map, as keygen takes them
                self.circuit_map(d_flags)
                d_flags.free()
            instance_cells = self.public_cells
        fc = vk["fixed_commitments"]
        doc = {"table_coeff": fixed[0:1], "cst_coeff": fixed[1:2], "sel_coeff": fixed[2:], "sigma_coeff": sigma, "lag_coeff": lag_coeff,
               "table_commits": fc[0:1], "cst_commits": fc[1:2], "sel_commits": fc[2:], "sigma_commits": vk["permutation_commitments"],
               "lag_commits": np.zeros((3, 8), dtype=np.uint64),
               "instance_cells": np.asarray(instance_cells, dtype=np.int64),
               "instance_cells_are_the_default": np.array([int(list(instance_cells) == list(self.public_cells))])}
        return self._install_key(doc)

    # ------------------------------------------------------------------ the rounds
    def prove(self, challenges=None, seed=None, timings=None, multiopen="shplonk", instances=None):
        """challenges: dict of Montgomery field elements beta, gamma, y, x, v, or None to derive them with the Fiat–Shamir
        transcript (api.Transcript; the proof bytes are then returned as `proof`).  Transcript order: the verifying key's
        digest (vk_digest: one scalar made of the fixed commitments at keygen); the public inputs; advice commitments -> theta (squeezed as halo2 does, unused: the lookups
        are single-column); permuted input / table commitments -> beta, gamma; product commitments, the vanishing argument's random
        polynomial -> y; the quotient's pieces -> x; all evaluations, rotation by rotation (not h's: h is folded at x and its value follows
        from the quotient identity, halo2's vanishing argument) -> then the multi-open: "gwc": v, one quotient per rotation
        point; "shplonk" (what the reference's gen_snark_shplonk runs, [UPSTREAM-RECALL] for the order of its challenges):
        yo, v; the quotient f of all rotation sets; u; the quotient of the linearisation polynomial.
        `instances`: the public values, one per public cell of keygen (Montgomery field elements); None = read from the witness
        this proof commits to (the honest prover's statement: circuit.instances(), src/scaffold/mod.rs:265).  They fill rows
        0 .. of the instance column, which the permutation argument ties to the public cells.
        Returns dict(commitments, evals, openings, points, proof, instances): commitments[name] (n, 8); evals[(name, rotation)] list
        of ints; openings: list of dict(rotation, point, polys=[names in combination order], eval, W).
        `seed`: None = every blinding scalar of this proof (advice, lookup and product columns) comes fresh from the
        operating system's entropy, as in halo2's create_proof; an integer makes the proof reproducible (tests).
        Sharded (world > 1, SHPLONK only): every rank calls prove() with the same arguments; each works on its own columns and
        sets and all end with the same proof bytes (see the class docstring and shardmap.py)."""
        hp, lib, rows, k, ne = self.hp, self.lib, self.rows, self.k, self.ne
        CHUNK_LEN, N_H, N_SLOTS = self.chunk_len, self.n_h, self.n_slots
        comm, world, rank = self.comm, self.world, self.rank
        if world > 1 and multiopen != "shplonk":
            raise ValueError("the sharded rounds open with SHPLONK")
        seeds = iter([None] * 9 if seed is None else [[int(seed), i] for i in range(1, 10)])
        tr = api.Transcript() if challenges is None else None
        ch = {} if challenges is None else {name: np.ascontiguousarray(v, dtype=np.uint64) for name, v in challenges.items()}
        p = {name: api._p(v) for name, v in ch.items()}

        host = self.host_ms = {"transcript": 0.0}     # wall-clock ms the host spent in the sponge (read by the benches)

        def squeeze(*names):
            if tr is not None:
                t0 = time.perf_counter()
                for name in names:
                    ch[name] = tr.squeeze()
                    p[name] = api._p(ch[name])
                host["transcript"] += (time.perf_counter() - t0) * 1e3

        def write_points(points):
            if tr is not None and len(points):
                t0 = time.perf_counter()
                tr.write_points(np.stack([np.asarray(pt) for pt in points]) if isinstance(points, list) else points)
                host["transcript"] += (time.perf_counter() - t0) * 1e3

        if tr is not None:
            tr.common_scalar(self.vk_digest())
        usable, n_adv, n_lk, n_cols, n_sets = self.usable, self.n_adv, self.n_lk, self.n_cols, self.n_sets
        a_lo, a_hi, l_lo, l_hi, my_adv, my_lk, my_sets = self.a_lo, self.a_hi, self.l_lo, self.l_hi, self.my_adv, self.my_lk, self.my_sets
        my_cols = my_adv + my_lk
        lk_ranges = [(l_lo, l_hi)]
        fx = self.fixed
        T = {} if timings is None else timings

        def stage(name, fn):
            # device time per stage only when asked for: the timer waits for the device, and an untimed proof lets the host's
            # transcript work run beside whatever the device still has queued
            if timings is None:
                return fn()
            api.timer_start()
            r = fn()
            T[name] = T.get(name, 0.0) + api.timer_stop()
            return r

        def power(name, e):
            """challenge^e as a Montgomery element (a fold that skips e terms another rank holds multiplies by it)"""
            return _fr_from_int(pow(_fr_to_int(ch[name]), int(e), R_MOD))

        class _Fold:
            """position in a sum  sum_i t_i c^(N-1-i)  that this rank folds its own terms of, in increasing i: `skip_to(i)`
            before term i is folded in (acc <- acc c + t_i) multiplies the accumulator by c for every term in between that
            another rank holds; `finish()` for the terms after its last.  On one rank nothing is ever skipped."""

            def __init__(self, acc, n_elems, challenge, n_terms):
                self.acc, self.n_elems, self.challenge, self.n_terms, self.pos, self.live = acc, n_elems, challenge, n_terms, 0, False

            def skip_to(self, i, count):
                if self.live and i > self.pos:
                    check(lib.vdb_poly_scale_dev(self.acc.ptr, api._p(power(self.challenge, i - self.pos)), _sz(self.n_elems)))
                assert i >= self.pos or not self.live, "terms are folded in increasing order"
                self.pos, self.live = i + count, True

            def finish(self):
                if self.live and self.n_terms > self.pos:
                    check(lib.vdb_poly_scale_dev(self.acc.ptr, api._p(power(self.challenge, self.n_terms - self.pos)), _sz(self.n_elems)))
                self.pos = self.n_terms

        # round 1: advice columns (the hot path of the bench: witness, commit, lagrange_to_coeff, coeff_to_extended)
        resident = hp.ext_cols >= my_cols + 2      # every advice coset stays in HBM; otherwise they are recomputed block by block below
        ni = len(self.instance_cells)
        given = None if instances is None else np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.uint64) for v in instances]).reshape(-1, 4) if ni else np.zeros((0, 4), np.uint64))
        assert given is None or len(given) == ni, "one value per public cell"
        inst_host = np.zeros((max(ni, 1), 4), dtype=np.uint64)

        def public_values():
            # rows 0 .. ni - 1 of the instance column <- the public cells of the witness just generated (or the values of the
            # statement handed in); the values are needed on the host before the advice commitments enter the transcript, so they
            # are read here, behind the witness kernels only, and not behind the transforms queued next
            if not ni:
                return
            if given is None:
                if world == 1:
                    check(lib.vdb_gather_fr_dev(hp.d_stream.ptr, self.d_inst_cells.ptr, _sz(ni), self.d_inst_lag.ptr))
                else:
                    # a rank writes only the cells of its own columns; the values every rank computes are the gadget's results
                    ptr, cnt = hp.public_values_dev()
                    if cnt != ni or self.instance_cells != self.public_cells:
                        raise ValueError("a sharded proof exposes the circuit's default public cells")
                    check(lib.vdb_memcpy_d2d(self.d_inst_lag.ptr, ptr, _sz(ni * B)))
                check(lib.vdb_memcpy_d2h(api._p(inst_host), self.d_inst_lag.ptr, _sz(ni * B)))
            else:
                inst_host[:ni] = given
                check(lib.vdb_memcpy_h2d(self.d_inst_lag.ptr, api._p(inst_host), _sz(ni * B)))
        # (the transforms of the advice columns are still running when step returns: the commitments are absorbed meanwhile)
        adv_local = hp.step(timings, blind_seed=None if seed is None else [int(seed), 0], with_ext=False, sync=False, after_witness=public_values).copy()
        if resident:        # the cosets of all my columns at once, into the hot path's coset buffer (three of the four it is sized for)
            stage("ntt", lambda: check(lib.vdb_coeff_to_cosets_dev(hp.d_cols.ptr, hp.d_ext.ptr, _sz(my_cols), k, N_SLOTS, None)))
        adv_commits = self._globalize(adv_local, self.adv_ranges, n_cols)
        instances = [inst_host[i].copy() for i in range(ni)]
        # the instance polynomial in the forms the rounds read (one column: queued behind the advice transforms)
        check(lib.vdb_memcpy_d2d(self.d_inst_coeff.ptr, self.d_inst_lag.ptr, _sz(rows * B)))
        check(lib.vdb_lagrange_to_coeff_dev(self.d_inst_coeff.ptr, _sz(1), k))
        check(lib.vdb_coeff_to_cosets_dev(self.d_inst_coeff.ptr, self.d_inst_ext.ptr, _sz(1), k, N_SLOTS, None))
        if resident:
            check(lib.vdb_memcpy_d2d(hp.d_ext.at((my_cols + 1) * ne * B), self.d_inst_ext.ptr, _sz(ne * B)))
        if tr is not None and ni:
            t0 = time.perf_counter()
            tr.common_scalars(inst_host[:ni])
            host["transcript"] += (time.perf_counter() - t0) * 1e3
        write_points(adv_commits)
        squeeze("theta")
        adv = _Poly("adv", my_cols, coeff=hp.d_cols, commits=adv_commits, ranges=self.adv_ranges, n_total=n_cols)
        n_perm = self.n_perm
        # the gate columns once more as a group of their own: only they are read at rows 1..3 (halo2 opens a column at the rotations
        # its queries name — the lookup columns only at the current row)
        polys = {"adv": adv, "advg": _Poly("advg", my_adv, coeff=hp.d_cols, commits=adv_commits[:n_adv], ranges=[(a_lo, a_hi)], n_total=n_adv)}
        # Everything below works on blocks of `blk` columns: the Lagrange forms, the sigma columns and every extended coset
        # exist one block at a time (the advice cosets too, unless the hot path keeps them resident); what stays in HBM is the
        # streams, the coefficient forms and the derived columns.
        blk = max(2 * CHUNK_LEN, min(self.block_cols, self.blk_alloc) // (2 * CHUNK_LEN) * (2 * CHUNK_LEN))
        d_lag_a, d_lag_s, d_ea, d_eb, d_ez, d_zf, d_zlast, d_lklag = self.d_lag_a, self.d_lag_s, self.d_ea, self.d_eb, self.d_ez, self.d_zf, self.d_zlast, self.d_lklag
        omega = api.root_of_unity(k)
        bp_p, n_bp = api._p(hp.bp), ctypes.c_uint64(len(hp.bp))
        from .pipeline import MINIMUM_ROWS
        lk_blind = hp.d_blind.at(hp.n_adv_cols * N_BLIND * B)
        foreign_slot = {c: i for i, c in enumerate(self.foreign)}

        def runs(c0, nb):
            """the permutation's columns c0 .. c0 + nb cut into stretches of one kind: (kind, first column, count) with kind
            "adv" / "lk" (this rank's), "foreign" (another rank's, received), "cst", "inst" """
            out = []
            for c in range(c0, c0 + nb):
                kind = ("cst" if c == n_cols else "inst" if c == n_cols + 1 else "adv" if a_lo <= c < a_hi else
                        "lk" if l_lo <= c - n_adv < l_hi else "foreign")
                if kind == "foreign" and c not in foreign_slot:
                    raise AssertionError("a column of this rank's sets that nobody sent")
                if out and out[-1][0] == kind and kind not in ("cst", "inst") and (kind != "foreign" or foreign_slot[c] == foreign_slot[c - 1] + 1):
                    out[-1][2] += 1
                else:
                    out.append([kind, c, 1])
            return out

        def lagrange_block(c0, nb, dest):
            """the permutation's columns c0 .. c0 + nb in Lagrange form: advice from the stream, lookup from their laid-out copy, constants, instances"""
            for kind, c, m in runs(c0, nb):
                to = dest.at((c - c0) * rows * B)
                if kind == "adv":
                    check(lib.vdb_layout_columns_range_dev(hp.d_stream.ptr, ctypes.c_uint64(hp.n_cells), bp_p, n_bp, k, ctypes.c_uint64(c), ctypes.c_uint64(c + m),
                                                           to, hp.d_blind.ptr, N_BLIND))
                else:
                    src = (d_lklag.at((c - n_adv - l_lo) * rows * B) if kind == "lk" else self.d_foreign_lag.at(foreign_slot[c] * rows * B) if kind == "foreign"
                           else fx["cst"].lag.ptr if kind == "cst" else self.d_inst_lag.ptr)
                    check(lib.vdb_memcpy_d2d(to, src, _sz(m * rows * B)))

        def local_ext_index(c):
            """position of permutation column c in the hot path's coset buffer [my advice | my lookup | constants | instance], None: not there"""
            loc = self._col_local(c)
            if loc is not None:
                return loc
            return my_cols + (c - n_cols) if c >= n_cols else None

        def adv_ext_block(c0, nb):
            """(pointer, first column) of a buffer that holds the cosets of the permutation's columns c0 .. c0 + nb"""
            if resident:
                loc = [local_ext_index(c) for c in range(c0, c0 + nb)]
                if None not in loc and loc == list(range(loc[0], loc[0] + nb)):
                    return hp.d_ext.ptr, c0 - loc[0]
            for kind, c, m in runs(c0, nb):
                to = d_ea.at((c - c0) * ne * B)
                if kind in ("adv", "lk"):
                    if resident:
                        check(lib.vdb_memcpy_d2d(to, hp.d_ext.at(self._col_local(c) * ne * B), _sz(m * ne * B)))
                    else:
                        check(lib.vdb_coeff_to_cosets_dev(hp.d_cols.at(self._col_local(c) * rows * B), to, _sz(m), k, N_SLOTS, None))
                elif kind == "foreign":
                    check(lib.vdb_coeff_to_cosets_dev(self.d_foreign_coeff.at(foreign_slot[c] * rows * B), to, _sz(m), k, N_SLOTS, None))
                else:
                    check(lib.vdb_memcpy_d2d(to, (fx["cst"].ext if kind == "cst" else self.d_inst_ext).ptr, _sz(ne * B)))
            return d_ea.ptr, c0

        def col_ptr(base, col0, c):
            return ctypes.c_void_p((base.value if hasattr(base, "value") else int(base)) + (c - col0) * ne * B)

        # round 2: the lookup argument's permuted columns
        counts = {"pa": my_lk, "ps": my_lk, "zp": my_sets, "zl": my_lk}
        der, off = {}, 0
        for name, m in counts.items():
            der[name] = _View(self.pool_der, off * rows * B, m * rows * B)
            off += m
        d_pa, d_ps, d_zp, d_zl = der["pa"], der["ps"], der["zp"], der["zl"]

        def permute():
            if my_lk:
                check(lib.vdb_layout_lookup_range_dev(hp.d_lookup.ptr, ctypes.c_uint64(hp.n_lookup), k, MINIMUM_ROWS, ctypes.c_uint64(l_lo), ctypes.c_uint64(l_hi),
                                                      d_lklag.ptr, lk_blind, N_BLIND))
            check(lib.vdb_lookup_permute_dev(d_lklag.ptr, fx["table"].lag.ptr, _sz(my_lk), _sz(rows), _sz(usable), hp.L, d_pa.ptr, d_ps.ptr))
        stage("lookup_permute", permute)
        self._blind(d_pa, my_lk, usable, next(seeds), lk_ranges, n_lk)
        self._blind(d_ps, my_lk, usable, next(seeds), lk_ranges, n_lk)
        pa_c = stage("commit_permuted", lambda: self._commit(d_pa, my_lk, 1, dense=False))
        ps_c = stage("commit_permuted", lambda: self._commit(d_ps, my_lk, 1, dense=False))
        if world > 1:
            both = self._globalize(np.concatenate([pa_c, ps_c], axis=1), lk_ranges, n_lk)
            pa_c, ps_c = np.ascontiguousarray(both[:, :8]), np.ascontiguousarray(both[:, 8:])
        polys["pa"] = _Poly("pa", my_lk, lag=d_pa, commits=pa_c, ranges=lk_ranges, n_total=n_lk)
        polys["ps"] = _Poly("ps", my_lk, lag=d_ps, commits=ps_c, ranges=lk_ranges, n_total=n_lk)
        if n_lk:
            write_points(np.stack([polys["pa"].commits, polys["ps"].commits], axis=1).reshape(-1, 8))     # (pa_c, ps_c) per lookup column
        squeeze("beta", "gamma")

        # the columns of my sets that another rank holds (the set that spans the advice / lookup junction): their holder sends the
        # Lagrange form, blinding rows included; the coefficient form is made here
        all_foreign = self.map.all_foreign_cols() if world > 1 else []
        if all_foreign:
            slab = np.zeros((len(all_foreign), rows, 4), dtype=np.uint64)
            tmp = api.DeviceBuffer(rows * B)
            for i, c in enumerate(all_foreign):
                if self._col_local(c) is not None:
                    lagrange_block(c, 1, tmp)
                    slab[i] = tmp.download((rows, 4))
            tmp.free()
            slab = comm.sum_disjoint(slab)
            for i, c in enumerate(all_foreign):
                if c in foreign_slot:
                    self.d_foreign_lag.upload(slab[i], offset=foreign_slot[c] * rows * B)
            if self.foreign:
                check(lib.vdb_memcpy_d2d(self.d_foreign_coeff.ptr, self.d_foreign_lag.ptr, _sz(len(self.foreign) * rows * B)))
                check(lib.vdb_lagrange_to_coeff_dev(self.d_foreign_coeff.ptr, _sz(len(self.foreign)), k))
            del slab

        # round 3 (beta, gamma): the running products of both arguments
        def products():
            for s_lo, s_hi in self.set_ranges:
                p_lo, p_hi = self.map.range_cols((s_lo, s_hi))
                for c0 in range(p_lo, p_hi, blk):
                    nb = min(blk, p_hi - c0)
                    lagrange_block(c0, nb, d_lag_a)
                    if self.d_map32 is not None:
                        check(lib.vdb_permutation_sigma_packed_dev(self.d_map32.at(c0 * rows * 4), _sz(nb), _sz(n_perm), k, api._p(self.delta), d_lag_s.ptr))
                    else:                                      # a key loaded from a file: back from the coefficient form
                        check(lib.vdb_memcpy_d2d(d_lag_s.ptr, fx["sigma"].coeff.at(self._sig_local(c0) * rows * B), _sz(nb * rows * B)))
                        check(lib.vdb_ntt_batch_dev(d_lag_s.ptr, _sz(nb), k, api._p(omega), 0))
                    check(lib.vdb_permutation_product_range_dev(d_lag_a.ptr, d_lag_s.ptr, _sz(nb), _sz(c0), k, _sz(usable), _sz(CHUNK_LEN), p["beta"], p["gamma"],
                                                                api._p(self.delta), d_zp.at(self._set_local(c0 // CHUNK_LEN) * rows * B)))
                # the products of a range run on from set to set
                check(lib.vdb_permutation_chain_dev(d_zp.at(self._set_local(s_lo) * rows * B), _sz(s_hi - s_lo), k, _sz(usable)))
            if world > 1:
                # ... and from range to range across the ranks: every range starts where the one before it ended.  One field
                # element per range is exchanged (its last running product at the last usable row), every rank multiplies up the
                # ranges before its own.
                order = self.map.all_ranges()
                ends = np.zeros((len(order), 4), dtype=np.uint64)
                for i, (lo, hi, r) in enumerate(order):
                    if r == rank:
                        check(lib.vdb_memcpy_d2h(api._p(ends[i]), d_zp.at((self._set_local(hi - 1) * rows + usable) * B), _sz(B)))
                ends = comm.sum_disjoint(ends)
                acc = 1
                for i, (lo, hi, r) in enumerate(order):
                    if r == rank and acc != 1:
                        check(lib.vdb_poly_scale_dev(d_zp.at(self._set_local(lo) * rows * B), api._p(_fr_from_int(acc)), _sz((hi - lo) * rows)))
                    acc = acc * _fr_to_int(ends[i]) % R_MOD
            check(lib.vdb_lookup_product_dev(d_lklag.ptr, fx["table"].lag.ptr, d_pa.ptr, d_ps.ptr, _sz(my_lk), _sz(rows), _sz(usable), p["beta"], p["gamma"],
                                             d_zl.ptr))
        stage("products", products)
        self._blind(d_zp, my_sets, usable + 1, next(seeds), self.set_ranges, n_sets)
        self._blind(d_zl, my_lk, usable + 1, next(seeds), lk_ranges, n_lk)
        def derived_forms(names):
            for name in names:
                q = polys[name]
                check(lib.vdb_lagrange_to_coeff_dev(q.lag.ptr, _sz(q.n_cols), k))      # in place: the Lagrange form is not needed again
                q.coeff, q.lag = q.lag, None
        # The host absorbs one batch of commitments while the device works on the next thing that needs no challenge: the lookup
        # products' MSM beside the permutation products' commitments, the coefficient forms (needed by the quotient, independent of y)
        # beside the lookup products' commitments.
        zp_c = self._globalize(stage("commit_products", lambda: self._commit(d_zp, my_sets, 1)), self.set_ranges, n_sets)
        polys["zp"] = _Poly("zp", my_sets, lag=d_zp, commits=zp_c, ranges=self.set_ranges, n_total=n_sets)
        stage("commit_products", lambda: self._commit_begin(d_zl, my_lk, 1))
        try:
            write_points(polys["zp"].commits)
        except Exception:
            lib.vdb_msm_batch_end(None, _sz(0))       # a deferred MSM must always be collected, or every later MSM is refused
            raise
        zl_c = self._globalize(stage("commit_products", lambda: self._commit_end(my_lk)), lk_ranges, n_lk)
        polys["zl"] = _Poly("zl", my_lk, lag=d_zl, commits=zl_c, ranges=lk_ranges, n_total=n_lk)
        stage("derived_ntt", lambda: derived_forms(("pa", "ps", "zp", "zl")))
        write_points(polys["zl"].commits)
        # The vanishing argument's random polynomial (halo2 plonk/vanishing/prover.rs Argument::commit, [UPSTREAM-RECALL]): n uniform
        # coefficients, committed before y is squeezed and opened at x beside the folded quotient — it blinds the one evaluation of h
        # the multi-open reveals.  In a sharded proof rank 0 draws it and every rank receives the same coefficients.
        d_rand = self.d_rand
        if rank == 0:
            api.random_scalars_dev(d_rand.ptr, rows, seed=next(seeds))
        if world > 1:
            rc = np.zeros((rows, 4), dtype=np.uint64)
            if rank == 0:
                check(lib.vdb_memcpy_d2h(api._p(rc), d_rand.ptr, _sz(rows * B)))
            rc = comm.sum_disjoint(rc)
            if rank != 0:
                d_rand.upload(rc)
        polys["rand"] = _Poly("rand", 1, coeff=d_rand, commits=stage("commit_h", lambda: self._commit(d_rand, 1, 0)), replicated=True)
        write_points(polys["rand"].commits)
        squeeze("y")

        # the boundary products other ranks ask for, in coefficient form: the set before each of their ranges, and the first set
        # for the rank that closes the chain
        zp = polys["zp"]
        z_slot = {i: j for j, i in enumerate(self.z_req)}
        all_z = self.map.all_z_requests() if world > 1 else []
        if all_z:
            slab = np.zeros((len(all_z), rows, 4), dtype=np.uint64)
            for j, i in enumerate(all_z):
                if self.map.set_owner(i) == rank:
                    check(lib.vdb_memcpy_d2h(api._p(slab[j]), zp.coeff.at(self._set_local(i) * rows * B), _sz(rows * B)))
            slab = comm.sum_disjoint(slab)
            for j, i in enumerate(all_z):
                if i in z_slot:
                    self.d_zhalo.upload(slab[j], offset=z_slot[i] * rows * B)
            del slab

        def z_coeff(i):
            """coefficient form of product polynomial i: mine, or the copy its owner sent"""
            if i in z_slot:
                return self.d_zhalo.at(z_slot[i] * rows * B)
            return zp.coeff.at(self._set_local(i) * rows * B)

        # round 4 (y): the quotient
        d_h = self.d_h
        l0, ll, la = (ctypes.c_void_p(fx["lag"].ext.ptr.value + i * ne * B) for i in range(3))

        def to_ext(coeff_ptr, dest_ptr, m):
            check(lib.vdb_coeff_to_cosets_dev(coeff_ptr, dest_ptr, _sz(m), k, N_SLOTS, None))

        def quotient():
            # The numerator is sum_i term_i y^(N-1-i) over the terms in halo2's order: gates (one per advice column),
            # the permutation argument (two terms of the product columns alone, the chaining of the sets, one product term per set),
            # the lookup argument (five per lookup column).  Each group is folded into an accumulator of its own (acc <- acc y +
            # term, from zero) while ONE sweep over blocks of columns produces every coset once — the advice / lookup / constants
            # cosets (unless resident), selectors, sigma, product and lookup-argument cosets — and the groups are joined at the end:
            # h = ((Ag y^n2 + A2) y^n3 + A3) y^n4 + A4.  (The public inputs have no term of their own: the instance column is one
            # of the permutation's columns.)
            # Everything is evaluated coset by coset on N_SLOTS = 3 of the extended domain's four cosets (the quotient has degree below
            # 3 n; 2 of 2 for a circuit of degree 3).  The gates have degree 3: their share of the quotient, Ag / (X^n - 1), has degree below 2 n, so Ag is evaluated on
            # two of them (slots 0 and 1 of the advice cosets; the selector cosets are made for those two only), brought back to
            # coefficients from there, and joined in coefficient form.
            # Sharded: a rank folds the terms of its own columns and sets (the folds skip what other ranks hold: _Fold); every step
            # after that — the joins, the division by X^n - 1, the way back to coefficients — is linear, so each rank ends with a
            # share of h's coefficients and the shares are added (the one bulk exchange of the proof: 2^(k+2) x 32 B per rank).
            ag, a2, a3, a4 = self.d_hg, self.d_h2, self.d_h3, self.d_h4
            for a in (a2, a3, a4):
                check(lib.vdb_memset_dev(a.ptr, 0, _sz(ne * B)))
            check(lib.vdb_memset_dev(ag.ptr, 0, _sz(GATE_SLOTS * rows * B)))
            n2, n3, n4 = 2 + (n_sets - 1), n_sets, 5 * n_lk
            f_g, f_2, f_3, f_4 = _Fold(ag, GATE_SLOTS * rows, "y", n_adv), _Fold(a2, ne, "y", n2), _Fold(a3, ne, "y", n3), _Fold(a4, ne, "y", n4)
            perm_args = (_sz(n_perm), _sz(CHUNK_LEN), k, N_SLOTS, _sz(usable), l0, ll, la, p["beta"], p["gamma"], api._p(self.delta), p["y"])
            # l0 (1 - z_0), l_last (z_last^2 - z_last): the first two terms of group 2, folded in by the owner of the last set
            if self.map.head_owner() == rank:
                to_ext(z_coeff(0), d_zf.ptr, 1)
                to_ext(z_coeff(n_sets - 1), d_zlast.ptr, 1)
                f_2.skip_to(0, 2)
                check(lib.vdb_permutation_eval_parts_cosets_dev(None, _sz(0), None, None, _sz(0), d_zf.ptr, d_zlast.ptr, *perm_args, a2.ptr, 1, _sz(0), _sz(0), _sz(0), _sz(0)))
            third = blk // 3
            sigma_scaled = os.environ.get("VDB_SIGMA_SCALED", "1") != "0"      # (0: the A/B baseline — sigma's own cosets, the product by beta per point)
            fixed_resident = getattr(self, "fixed_cosets_resident", False)

            def lookup_terms(base, col0, j_lo, j_hi):
                """the lookup argument of my lookup columns j_lo .. j_hi, whose input cosets are in the block at `base`:
                [permuted input | permuted table | product] cosets in thirds of one buffer"""
                for j0 in range(j_lo, max(j_hi, j_lo), third):
                    m = min(third, j_hi - j0)
                    to_ext(polys["pa"].coeff.at((j0 - l_lo) * rows * B), d_eb.ptr, m)
                    to_ext(polys["ps"].coeff.at((j0 - l_lo) * rows * B), d_eb.at(third * ne * B), m)
                    to_ext(polys["zl"].coeff.at((j0 - l_lo) * rows * B), d_eb.at(2 * third * ne * B), m)
                    f_4.skip_to(5 * j0, 5 * m)
                    check(lib.vdb_lookup_eval_cosets_dev(col_ptr(base, col0, n_adv + j0), fx["table"].ext.ptr, d_eb.ptr, d_eb.at(third * ne * B), d_eb.at(2 * third * ne * B),
                                                         _sz(m), k, N_SLOTS, l0, ll, la, p["beta"], p["gamma"], p["y"], a4.ptr))
            # my lookup columns that complete another rank's set (the head of the lookup columns, at the advice / lookup junction):
            # their lookup argument is mine all the same, and comes first in the order of the terms
            stray_lk = sorted(c for c in self.stray if c >= n_adv)
            if stray_lk:
                assert stray_lk == list(range(stray_lk[0], stray_lk[0] + len(stray_lk)))
                base, col0 = adv_ext_block(stray_lk[0], len(stray_lk))
                lookup_terms(base, col0, stray_lk[0] - n_adv, stray_lk[-1] + 1 - n_adv)
            assert all(c >= n_adv for c in self.stray), "an advice column outside its rank's sets"
            for s_lo, s_hi in self.set_ranges:
                p_lo, p_hi = self.map.range_cols((s_lo, s_hi))
                for c0 in range(p_lo, p_hi, blk):
                    nb = min(blk, p_hi - c0)
                    base, col0 = adv_ext_block(c0, nb)
                    g0, g1 = max(c0, a_lo), min(c0 + nb, a_hi)
                    if g0 < g1:                                            # gates of the block's advice columns
                        if fixed_resident:
                            sel_ptr = fx["sel"].ext.at((g0 - a_lo) * GATE_SLOTS * rows * B)
                        else:
                            check(lib.vdb_coeff_to_cosets_dev(fx["sel"].coeff.at((g0 - a_lo) * rows * B), d_eb.ptr, _sz(g1 - g0), k, GATE_SLOTS, None))
                            sel_ptr = d_eb.ptr
                        f_g.skip_to(g0, g1 - g0)
                        check(lib.vdb_gate_eval_cosets_dev(col_ptr(base, col0, g0), N_SLOTS, sel_ptr, _sz(g1 - g0), k, GATE_SLOTS, p["y"], ag.ptr))
                    # permutation: the block's sets with their sigma cosets and product cosets (one set more in front for the chaining)
                    set_lo, set_hi = c0 // CHUNK_LEN, -(-(c0 + nb) // CHUNK_LEN)
                    z0 = max(set_lo - 1, 0)
                    # (the cosets of beta sigma: the scalar rides on the transform's coset factors, the evaluation skips a product per point)
                    # resident sigma cosets (keygen): no transform, the evaluation multiplies by beta itself
                    if fixed_resident:
                        sig_ptr, sig_head = fx["sigma"].ext.at(self._sig_local(c0) * ne * B), 0
                    elif sigma_scaled:
                        check(lib.vdb_coeff_to_cosets_dev(fx["sigma"].coeff.at(self._sig_local(c0) * rows * B), d_eb.ptr, _sz(nb), k, N_SLOTS, p["beta"]))
                        sig_ptr, sig_head = d_eb.ptr, 2
                    else:
                        to_ext(fx["sigma"].coeff.at(self._sig_local(c0) * rows * B), d_eb.ptr, nb)
                        sig_ptr, sig_head = d_eb.ptr, 0
                    if z0 < set_lo and (z0 in z_slot or z0 < s_lo):          # the set in front is another rank's (or another range's)
                        to_ext(z_coeff(z0), d_ez.ptr, 1)
                        to_ext(z_coeff(set_lo), d_ez.at(ne * B), set_hi - set_lo)
                    else:
                        to_ext(z_coeff(z0), d_ez.ptr, set_hi - z0)
                    if max(set_lo, 1) < set_hi:
                        f_2.skip_to(max(set_lo, 1) + 1, set_hi - max(set_lo, 1))
                        check(lib.vdb_permutation_eval_parts_cosets_dev(None, _sz(0), None, d_ez.ptr, _sz(z0), None, None, *perm_args, a2.ptr, 0, _sz(max(set_lo, 1)), _sz(set_hi),
                                                                 _sz(0), _sz(0)))
                    f_3.skip_to(set_lo, set_hi - set_lo)
                    check(lib.vdb_permutation_eval_parts_cosets_dev(base, _sz(col0), sig_ptr, d_ez.ptr, _sz(z0), None, None, *perm_args, a3.ptr, sig_head, _sz(0), _sz(0), _sz(set_lo),
                                                             _sz(set_hi)))
                    # lookup argument of the block's lookup columns that are mine
                    j_lo, j_hi = max(c0 - n_adv, l_lo), min(c0 + nb - n_adv, l_hi)
                    lookup_terms(base, col0, j_lo, j_hi)
            for f in (f_g, f_2, f_3, f_4):
                f.finish()
            # join the groups that live on the 4 n points (acc_next += y^(terms of the next group) * acc), divide, back to coefficients
            y_int = _fr_to_int(ch["y"])
            check(lib.vdb_poly_axpy_dev(a3.ptr, api._p(_fr_from_int(pow(y_int, n3, R_MOD))), a2.ptr, _sz(ne)))
            check(lib.vdb_poly_axpy_dev(a4.ptr, api._p(_fr_from_int(pow(y_int, n4, R_MOD))), a3.ptr, _sz(ne)))
            # (division by X^n - 1 — a constant per coset —, residues modulo X^n - g_t^n, the Vandermonde system in g_t^n: N_H pieces)
            check(lib.vdb_cosets_to_coeff_dev(a4.ptr, d_h.ptr, k, N_SLOTS))
            # the gates' share: the same from its two cosets, then h += y^(every later term) * (its 2 n coefficients)
            check(lib.vdb_cosets_to_coeff_dev(ag.ptr, a2.ptr, k, GATE_SLOTS))
            check(lib.vdb_poly_axpy_dev(d_h.ptr, api._p(_fr_from_int(pow(y_int, n2 + n3 + n4, R_MOD))), a2.ptr, _sz(GATE_SLOTS * rows)))
            comm.sum_field_dev(d_h.ptr, ne)                      # every rank's share of h (nothing to do on one rank)
        stage("quotient", quotient)
        n_h = N_H                                             # h(X) = sum_i X^(n i) h_i(X), degree below (degree - 1) n
        polys["h"] = _Poly("h", n_h, coeff=d_h, commits=stage("commit_h", lambda: self._commit(d_h, n_h, 0)), replicated=True)
        write_points(polys["h"].commits)
        squeeze("x")
        # h folded at x (halo2 vanishing::Constructed::evaluate): hf(X) = sum_i x^(n i) h_i(X), one polynomial of degree below n whose
        # commitment the verifier forms from the pieces' and whose value at x it computes from the quotient identity — neither is sent
        xn = pow(_fr_to_int(ch["x"]), rows, R_MOD)
        d_hf = self.d_hf
        check(lib.vdb_memcpy_d2d(d_hf.ptr, d_h.ptr, _sz(rows * B)))
        for i in range(1, n_h):
            check(lib.vdb_poly_axpy_dev(d_hf.ptr, api._p(_fr_from_int(pow(xn, i, R_MOD))), d_h.at(i * rows * B), _sz(rows)))
        polys["hf"] = _Poly("hf", 1, coeff=d_hf, commits=stage("commit_h", lambda: self._commit(d_hf, 1, 0)), replicated=True)

        # round 5 (x): evaluations.  Which polynomial is read at which rotation: the gate reads the advice at rows 0..3, the
        # products one row ahead, the permuted input one row back, the chained product N_BLIND rows back.
        allp = {**polys, **fx}
        opened = {0: ["adv", "sel", "sigma", "cst", "table", "pa", "ps", "zp", "zl", "hf", "rand"], 1: ["advg", "zp", "zl"], 2: ["advg"], 3: ["advg"], -1: ["pa"],
                  -N_BLIND: ["zp"]}
        opened = {rot: [name for name in names if allp[name].n_total] for rot, names in opened.items()}    # a circuit without lookups
        opened = {rot: names for rot, names in opened.items() if names}                                   # opens nothing at w^-1 x
        x_int = _fr_to_int(ch["x"])
        w_int = _fr_to_int(api.root_of_unity(k))
        evals, points = {}, {}

        groups = [(rot, name) for rot, names in opened.items() for name in names]
        for rot in opened:
            points[rot] = x_int * pow(w_int, rot % rows, R_MOD) % R_MOD

        def evaluate():
            for rot, name in groups:
                q = allp[name]
                out = np.zeros((q.n_cols, 4), dtype=np.uint64)
                check(lib.vdb_eval_polys_dev(q.coeff.ptr, _sz(q.n_cols), _sz(rows), api._p(_fr_from_int(points[rot])), api._p(out)))
                evals[(name, rot)] = out
            if world > 1:
                # every rank evaluated its own polynomials: one exchange puts every group's evaluations in the global order
                # (the polynomials every rank holds are not exchanged)
                shared = [(rot, name) for rot, name in groups if not allp[name].replicated]
                whole = np.zeros((sum(allp[name].n_total for _rot, name in shared), 4), dtype=np.uint64)
                o = 0
                for rot, name in shared:
                    q, loc = allp[name], 0
                    for lo, hi in q.ranges:
                        whole[o + lo: o + hi] = evals[(name, rot)][loc: loc + hi - lo]
                        loc += hi - lo
                    o += q.n_total
                whole = comm.sum_disjoint(whole)
                o = 0
                for rot, name in shared:
                    evals[(name, rot)] = np.ascontiguousarray(whole[o: o + allp[name].n_total])
                    o += allp[name].n_total

        def evaluate_and_absorb():
            # the device evaluates group i + 1 while the host absorbs the evaluations of group i (the sponge's host work — ~6 us per four
            # values — is longer than the evaluation itself: only the first group's kernel is not hidden)
            total = sum(allp[name].n_cols for _rot, name in groups)
            d_ev = api.DeviceBuffer(max(total, 1) * B)
            offs, o = [], 0
            for _rot, name in groups:
                offs.append(o)
                o += allp[name].n_cols * B

            def launch(i):
                rot, name = groups[i]
                q = allp[name]
                check(lib.vdb_eval_polys_dev_out(q.coeff.ptr, _sz(q.n_cols), _sz(rows), api._p(_fr_from_int(points[rot])), d_ev.at(offs[i])))
            try:
                if groups:
                    launch(0)
                for i, (rot, name) in enumerate(groups):
                    out = d_ev.download((allp[name].n_cols, 4), offset=offs[i])       # waits for group i's kernel only
                    if i + 1 < len(groups):
                        launch(i + 1)
                    evals[(name, rot)] = out
                    if name in DERIVED:                  # computed by the verifier, not sent
                        continue
                    t0 = time.perf_counter()
                    tr.write_scalars(out)
                    tr.flush()
                    host["transcript"] += (time.perf_counter() - t0) * 1e3
            finally:
                api.sync()
                d_ev.free()

        if world == 1 and tr is not None and timings is None and os.environ.get("VDB_EVAL_PIPELINE", "1") != "0":
            evaluate_and_absorb()
        else:
            stage("evaluations", evaluate)
            if tr is not None:
                t0 = time.perf_counter()
                for rot, name in groups:
                    if name not in DERIVED:
                        tr.write_scalars(evals[(name, rot)])
                host["transcript"] += (time.perf_counter() - t0) * 1e3
        d_comb, d_quot = self.d_comb, self.d_quot
        if multiopen == "shplonk":
            openings = self._shplonk(allp, opened, points, evals, p, ch, squeeze, write_points, stage, power)
            proof = None
            if tr is not None:
                proof = tr.proof()
                tr.free()
            api.sync()
            return dict(commitments={name: q.commits for name, q in allp.items()}, evals=openings.pop("evals_int"), openings=openings, points=points,
                        proof=proof, challenges={name: v.copy() for name, v in ch.items()}, opened=opened,
                        instances=[_fr_to_int(v) for v in instances])
        squeeze("v")

        # round 6 (v): one opening per rotation point: combine with powers of v, divide by (X - point), commit
        openings = []

        def open_all():
            for rot, names in opened.items():
                check(lib.vdb_memset_dev(d_comb.ptr, 0, _sz(rows * B)))
                for name in names:
                    q = allp[name]
                    check(lib.vdb_poly_lincomb_dev(q.coeff.ptr, _sz(q.n_cols), _sz(rows), p["v"], d_comb.ptr))
                ptm = _fr_from_int(points[rot])
                rem = np.zeros((1, 4), dtype=np.uint64)
                check(lib.vdb_kate_div_dev(d_comb.ptr, _sz(1), _sz(rows), api._p(ptm), d_quot.ptr, api._p(rem)))
                W = self._commit(d_quot, 1, 0)[0]
                openings.append(dict(rotation=rot, point=points[rot], polys=list(names), eval=rem[0].copy(), W=W))
        stage("openings", open_all)
        write_points([op["W"] for op in openings])
        proof = None
        if tr is not None:
            proof = tr.proof()
            tr.free()
        api.sync()
        commitments = {name: q.commits for name, q in allp.items()}
        return dict(commitments=commitments, evals=_Evals(evals), openings=openings, points=points,
                    proof=proof, challenges={name: v.copy() for name, v in ch.items()}, opened=opened,
                        instances=[_fr_to_int(v) for v in instances])

    # ------------------------------------------------------------------ SHPLONK multi-open (halo2 poly/kzg/multiopen/shplonk)
    def _shplonk(self, allp, opened, points, evals, p, ch, squeeze, write_points, stage, power):
        """Polynomials opened at the same set of points form a rotation set S.  With q_S = the set's polynomials combined with
        powers of yo, r_S the interpolant of q_S's values on S and Z_S the vanishing polynomial of S:
            f = sum_S v^(m-1-s) (q_S - r_S) / Z_S                          -> commitment W1, then u,
            L = sum_S v^(m-1-s) Z_{T minus S}(u) (q_S - r_S(u)) - Z_T(u) f,  L(u) = 0  -> W2 = commit(L / (X - u)).
        The polynomial work (combinations, divisions by the linear factors, scaled sums, commits) runs on the device; the
        interpolants have at most four points and are host integers.
        Sharded: q_S is a sum over the set's polynomials, so each rank combines the ones it holds (their powers of yo by their
        place in the whole set); the quotient of a division by Z_S — the remainder dropped — and L's division by X - u are
        linear, so every rank commits its share of f and of L / (X - u) and the shares are added as points (vdb_g1_sum: RCCL has no
        curve operator, two 64-byte points per rank are gathered)."""
        lib, rows = self.lib, self.rows
        comm, world, rank = self.comm, self.world, self.rank
        R = R_MOD
        by_poly = {}
        for rot, names in opened.items():
            for name in names:
                by_poly.setdefault(name, []).append(rot)
        sets = []
        for name, rots in by_poly.items():
            key = tuple(sorted(rots))
            for sset in sets:
                if sset[0] == key:
                    sset[1].append(name)
                    break
            else:
                sets.append((key, [name]))
        squeeze("yo", "v")
        yo, v = _fr_to_int(ch["yo"]), _fr_to_int(ch["v"])
        m = len(sets)
        d_q = [api.DeviceBuffer(rows * B) for _ in sets]
        d_f, d_a, d_b = api.DeviceBuffer(rows * B), self.d_comb, self.d_quot
        ev_int = _Evals(evals)

        def interpolate(pts, vals):          # coefficients (low first) of the polynomial through (pts, vals)
            coeffs = [0] * len(pts)
            for i, (xi, yi) in enumerate(zip(pts, vals)):
                basis, denom = [1], 1
                for j, xj in enumerate(pts):
                    if j != i:
                        basis = [(a - xj * b) % R for a, b in zip([0] + basis, basis + [0])]
                        denom = denom * (xi - xj) % R
                scale = yi * pow(denom, -1, R) % R
                coeffs = [(c + scale * b) % R for c, b in zip(coeffs, basis)]
            return coeffs

        def at(coeffs, x):
            acc = 0
            for c in reversed(coeffs):
                acc = (acc * x + c) % R
            return acc

        def vanish(rots, x):
            acc = 1
            for rot in rots:
                acc = acc * (x - points[rot]) % R
            return acc

        def combine(names, dest):
            """dest <- this rank's share of the set's polynomials combined with powers of yo: polynomial j of the M the set has in
            all enters with yo^(M - 1 - j).  Horner over the ones held here; a stretch another rank holds multiplies by yo^(its length)."""
            check(lib.vdb_memset_dev(dest.ptr, 0, _sz(rows * B)))
            pos, base, live = 0, 0, False
            for name in names:
                q = allp[name]
                loc = 0
                for lo, hi in (q.ranges if (not q.replicated or rank == 0) else []):
                    if live and base + lo > pos:
                        check(lib.vdb_poly_scale_dev(dest.ptr, api._p(power("yo", base + lo - pos)), _sz(rows)))
                    check(lib.vdb_poly_lincomb_dev(q.coeff.at(loc * rows * B), _sz(hi - lo), _sz(rows), p["yo"], dest.ptr))
                    pos, live, loc = base + hi, True, loc + hi - lo
                base += q.n_total
            if live and base > pos:
                check(lib.vdb_poly_scale_dev(dest.ptr, api._p(power("yo", base - pos)), _sz(rows)))

        def share(point):
            """the sum over the ranks of their partial commitments"""
            if world == 1:
                return point
            out = np.zeros((1, 8), dtype=np.uint64)
            parts = np.ascontiguousarray(comm.gather_rows(np.asarray(point, dtype=np.uint64).reshape(1, 8)))
            check(lib.vdb_g1_sum(api._p(parts), _sz(world), _sz(1), api._p(out)))
            return out[0]

        r_polys, rems = [], []

        def quotient_f():
            check(lib.vdb_memset_dev(d_f.ptr, 0, _sz(rows * B)))
            for s_i, (rots, names) in enumerate(sets):
                combine(names, d_q[s_i])
                vals = []
                for rot in rots:                                   # the set's evaluations at this point, combined with powers of yo (host, compiled)
                    acc = np.zeros(4, dtype=np.uint64)
                    for name in names:
                        e = np.ascontiguousarray(evals[(name, rot)], dtype=np.uint64)
                        check(lib.vdb_fr_horner(api._p(e), _sz(e.shape[0]), p["yo"], api._p(acc)))
                    vals.append(_fr_to_int(acc))
                r = interpolate([points[rot] for rot in rots], vals)
                r_polys.append(r)
                # (q_S - r_S) / Z_S: the low coefficients on the host, one division per point on the device.  (r_S has fewer
                # coefficients than Z_S has roots: it changes the remainders only, which is why a rank's share needs no r_S.)
                check(lib.vdb_memcpy_d2d(d_a.ptr, d_q[s_i].ptr, _sz(rows * B)))
                if world == 1:
                    low = np.zeros((len(r), 4), dtype=np.uint64)
                    check(lib.vdb_memcpy_d2h(api._p(low), d_a.ptr, _sz(low.nbytes)))
                    low = np.stack([_fr_from_int(_fr_to_int(low[i]) - r[i]) for i in range(len(r))])
                    check(lib.vdb_memcpy_h2d(d_a.ptr, api._p(low), _sz(low.nbytes)))
                src, dst = d_a, d_b
                for rot in rots:
                    rem = np.zeros((1, 4), dtype=np.uint64)
                    check(lib.vdb_kate_div_dev(src.ptr, _sz(1), _sz(rows), api._p(_fr_from_int(points[rot])), dst.ptr, api._p(rem)))
                    rems.append(_fr_to_int(rem[0]))
                    src, dst = dst, src
                check(lib.vdb_poly_lincomb_dev(src.ptr, _sz(1), _sz(rows), p["v"], d_f.ptr))          # f = f v + (q_S - r_S) / Z_S
            return share(self._commit(d_f, 1, 0)[0])
        W1 = stage("openings", quotient_f)
        write_points([W1])
        squeeze("u")
        u = _fr_to_int(ch["u"])
        all_rots = sorted({rot for rots, _ in sets for rot in rots})

        def linearisation():
            check(lib.vdb_memset_dev(d_a.ptr, 0, _sz(rows * B)))
            const = 0
            for s_i, (rots, names) in enumerate(sets):
                coef = pow(v, m - 1 - s_i, R) * vanish([rot for rot in all_rots if rot not in rots], u) % R
                check(lib.vdb_poly_axpy_dev(d_a.ptr, api._p(_fr_from_int(coef)), d_q[s_i].ptr, _sz(rows)))
                const = (const + coef * at(r_polys[s_i], u)) % R
            check(lib.vdb_poly_axpy_dev(d_a.ptr, api._p(_fr_from_int(-vanish(all_rots, u))), d_f.ptr, _sz(rows)))
            if rank == 0:                                     # the constant term belongs to one share
                c0 = np.zeros((1, 4), dtype=np.uint64)
                check(lib.vdb_memcpy_d2h(api._p(c0), d_a.ptr, _sz(32)))
                c0[0] = _fr_from_int(_fr_to_int(c0[0]) - const)
                check(lib.vdb_memcpy_h2d(d_a.ptr, api._p(c0), _sz(32)))
            rem = np.zeros((1, 4), dtype=np.uint64)
            check(lib.vdb_kate_div_dev(d_a.ptr, _sz(1), _sz(rows), p["u"], d_b.ptr, api._p(rem)))
            rems.append(_fr_to_int(rem[0]))
            return share(self._commit(d_b, 1, 0)[0])
        W2 = stage("openings", linearisation)
        write_points([W2])
        for b in d_q + [d_f]:
            b.free()
        return dict(kind="shplonk", sets=[(list(rots), list(names)) for rots, names in sets], W1=W1, W2=W2, remainders=rems, evals_int=ev_int)

    def free(self):
        for q in self.fixed.values():
            q.free()
        self.fixed = {}
        self._vk_digest = None
        for name in ("pool_der", "d_lklag", "d_lag_a", "d_lag_s", "d_ea", "d_eb", "d_ez", "d_zf", "d_zlast", "d_h", "d_h2", "d_h3", "d_h4", "d_hg", "d_comb", "d_quot", "d_map32", "d_inst_lag", "d_inst_coeff", "d_inst_ext", "d_inst_cells",
                     "d_foreign_lag", "d_foreign_coeff", "d_zhalo", "d_rand", "d_hf", "d_key"):
            b = getattr(self, name, None)
            if b is not None:
                b.free()
                setattr(self, name, None)
        if hasattr(getattr(self, "circuit", None), "free"):
            self.circuit.free()
        for name in ("srs_m", "srs_few"):
            if getattr(self, name, None) is not None:
                getattr(self, name).free()
                setattr(self, name, None)


def instance_eval(instances, x, k):
    """The instance polynomial at x from the public values alone (what a verifier does instead of reading an evaluation from the
    proof): sum_i v_i L_i(x), L_i(x) = w^i (x^n - 1) / (n (x - w^i)) over the domain of 2^k rows.  Canonical integers."""
    n = 1 << k
    w = _fr_to_int(api.root_of_unity(k))
    if not instances:
        return 0
    xn1 = (pow(x, n, R_MOD) - 1) % R_MOD
    acc, wi = 0, 1
    dens = []
    for _ in instances:
        dens.append((x - wi) % R_MOD)
        wi = wi * w % R_MOD
    # one inversion for all denominators
    pref = [1]
    for d in dens:
        pref.append(pref[-1] * d % R_MOD)
    inv = pow(pref[-1], -1, R_MOD)
    wi_list = [pow(w, i, R_MOD) for i in range(len(instances))]
    for i in range(len(instances) - 1, -1, -1):
        di = inv * pref[i] % R_MOD
        inv = inv * dens[i] % R_MOD
        acc = (acc + int(instances[i]) * wi_list[i] % R_MOD * di) % R_MOD
    return acc * xn1 % R_MOD * pow(n, -1, R_MOD) % R_MOD


def lagrange_evals(x, k, usable):
    """(l_0(x), l_last(x), l_active(x)) as a verifier computes them (halo2 EvaluationDomain::l_i_range): l_i(x) = w^i (x^n - 1) / (n (x - w^i));
    l_last = l_usable, l_active = 1 - l_last - l_blind with l_blind the sum over the rows behind `usable`"""
    R, n = R_MOD, 1 << k
    w = _fr_to_int(api.root_of_unity(k))
    zn = (pow(x, n, R) - 1) * pow(n, -1, R) % R
    li = lambda i: pow(w, i, R) * zn % R * pow((x - pow(w, i, R)) % R, -1, R) % R
    l_last = li(usable)
    l_blind = sum(li(i) for i in range(usable + 1, n)) % R
    return li(0), l_last, (1 - l_last - l_blind) % R


def quotient_identity_holds(pr, challenges, evals, instances=None):
    """What a verifier checks first: the gate, permutation and lookup expressions recombined from the evaluations at x (and the
    rotated points) equal h(x) (x^n - 1).  `pr`: the ProverRounds that produced them (for the circuit's shape); `challenges`,
    `evals`: as returned by ProverRounds.prove.  Plain integer arithmetic on the host."""
    R = R_MOD
    b, g, yv, x = (_fr_to_int(challenges[name]) for name in ("beta", "gamma", "y", "x"))
    delta, n, n_adv = _fr_to_int(pr.delta), pr.rows, pr.n_adv
    ev = lambda name, rot=0: evals.get((name, rot), [])
    acc = 0
    a0, a1, a2, a3, q = ev("adv"), ev("advg", 1), ev("advg", 2), ev("advg", 3), ev("sel")
    for c in range(n_adv):
        acc = (acc * yv + q[c] * (a0[c] + a1[c] * a2[c] - a3[c])) % R
    l0, ll, la = lagrange_evals(x, pr.k, pr.usable)
    sg, z0, z1, zb = ev("sigma"), ev("zp"), ev("zp", 1), ev("zp", -N_BLIND)
    # the permutation's columns: advice, lookup, the constants' fixed column, the instance column (evaluated from the public values)
    pcols = list(a0) + list(ev("cst")) + [instance_eval(list(instances) if instances is not None else [], x, pr.k)]
    n_cols, n_sets = len(pcols), len(z0)
    acc = (acc * yv + l0 * (1 - z0[0])) % R
    acc = (acc * yv + ll * (z0[-1] * z0[-1] - z0[-1])) % R
    for i in range(1, n_sets):
        acc = (acc * yv + l0 * (z0[i] - zb[i - 1])) % R
    cur = b * x % R
    for i in range(n_sets):
        left, right = z1[i], z0[i]
        for c in range(i * pr.chunk_len, min((i + 1) * pr.chunk_len, n_cols)):
            left = left * (pcols[c] + b * sg[c] + g) % R
            right = right * (pcols[c] + cur + g) % R
            cur = cur * delta % R
        acc = (acc * yv + la * (left - right)) % R
    A, S, PA, PS, PAm, Z, Z1 = a0[n_adv:], ev("table")[0], ev("pa"), ev("ps"), ev("pa", -1), ev("zl"), ev("zl", 1)
    for c in range(len(PA)):
        acc = (acc * yv + l0 * (1 - Z[c])) % R
        acc = (acc * yv + ll * (Z[c] * Z[c] - Z[c])) % R
        acc = (acc * yv + la * (Z1[c] * (PA[c] + b) * (PS[c] + g) - Z[c] * (A[c] + b) * (S + g))) % R
        acc = (acc * yv + l0 * (PA[c] - PS[c])) % R
        acc = (acc * yv + la * (PA[c] - PS[c]) * (PA[c] - PAm[c])) % R
    xn = pow(x, n, R)
    hx = ev("hf")[0]                 # h folded at x: sum_i x^(n i) h_i(x), evaluated by the prover (a verifier computes it from this very identity)
    return acc == hx * (xn - 1) % R and acc != 0
