"""HBM-resident driver of the proving hot path for one circuit (plumbing above the C ABI).

Mirrors what the reference's `Prove` arm does up to the committed / extended advice columns
(/root/reference/src/scaffold/mod.rs:284-298 -> create_proof):

    closure (witness gen)  ->  assign_threads_in (stream -> columns)  ->  commit_lagrange per column
    ->  lagrange_to_coeff  ->  coeff_to_extended

`setup()` plays the Keygen arm's role for the layout (src/scaffold/mod.rs:267-283): one run with gate
selectors recorded, from which the break points ("pinning") are derived.  Everything stays in HBM;
only commitments (64 B per column) come back to the host.
"""
import ctypes
import math

import numpy as np

from . import api
from ._lib import check

MINIMUM_ROWS = 9   # MINIMUM_ROWS default, src/scaffold/mod.rs:383
# Rows at the end of every column that the prover fills with random scalars: halo2's blinding_factors() + 1.  blinding_factors() =
# max(3, advice queries per column) + 2 [UPSTREAM-RECALL]; halo2-base's vertical gate reads a column at four rotations, so 6 — which is
# what the reference's own MINIMUM_ROWS = 9 = blinding_factors() + 3 says (src/scaffold/mod.rs:383) — and the last usable row, where
# l_last sits and the running products end, is row 2^k - 7.  (Rounds 1-2 used 6.)
N_BLIND = 7


def sift_like_vectors(seed, n, dim, k_distinct=0):
    """integers uniform in [0, 218] stored as f64; rows non-zero; first K rows distinct."""
    while True:
        rng = np.random.default_rng(seed)
        v = rng.integers(0, 219, size=(n, dim)).astype(np.float64)
        ok = (v.sum(axis=1) > 0).all()
        if k_distinct:
            ok = ok and len({tuple(r) for r in v[:k_distinct]}) == k_distinct
        if ok:
            return v, seed
        seed += 1


def shard_range(n_cols, rank, world):
    """Contiguous block [lo, hi) of a list of `n_cols` columns owned by `rank`."""
    return n_cols * rank // world, n_cols * (rank + 1) // world


def column_shards(n_adv, n_lk, world):
    """Per rank: (advice block, lookup block).  Advice and lookup columns are split separately so that every rank's
    blocks cover about the same stretch of the witness stream (lookup cells are emitted alongside the advice cells
    that are range-checked), which is what lets each rank generate only its part of the witness."""
    return [(shard_range(n_adv, r, world), shard_range(n_lk, r, world)) for r in range(world)]


def balanced_ranges(cost, world):
    """Contiguous blocks of a column list with (nearly) equal summed cost: block r ends at the first column where the
    running cost reaches (r + 1) / world of the total."""
    cost = np.asarray(cost, dtype=np.float64)
    n = len(cost)
    if n == 0:
        return [(0, 0)] * world
    cum = np.cumsum(cost)
    cuts = [0]
    for r in range(1, world):
        cuts.append(max(cuts[-1], min(n, int(np.searchsorted(cum, cum[-1] * r / world, side="left")) + 1)))
    cuts.append(n)
    return [(cuts[r], max(cuts[r], cuts[r + 1])) for r in range(world)]


def balanced_column_shards(var_adv, var_lk, world, msm_share=0.22):
    """Like column_shards, but the blocks equalise estimated time instead of column count: every column costs one unit
    (NTT, layout, bucket reduction) plus `msm_share` units per mean-column's worth of MSM entries (the non-zero signed
    window digits of its non-constant cells: what its commitment sorts and accumulates; the share is the measured ratio
    of those two groups of kernels: ~9 us of accumulate / combine per mean column against ~33 us of NTT, sort and
    bucket reduction).  Witness columns differ a lot here (input and indicator columns are
    nearly empty, fixed-point columns hold 100-bit values), which is what skews equal-count blocks."""
    var_adv = np.asarray(var_adv, dtype=np.float64)
    var_lk = np.asarray(var_lk, dtype=np.float64)
    mean = max(1.0, (var_adv.sum() + var_lk.sum()) / max(1, len(var_adv) + len(var_lk)))
    adv = balanced_ranges(1.0 + msm_share * var_adv / mean, world)
    lk = balanced_ranges(1.0 + msm_share * var_lk / mean, world)
    return list(zip(adv, lk))


def set_cols(n_lk_cols):
    """columns per product polynomial of the permutation argument, halo2's chunk_len = cs.degree() - 2 (rounds.constraint_degree: 4 for a
    circuit with lookup columns, 3 without): block boundaries of a sharded job fall on sets"""
    return 2 if n_lk_cols else 1


SET_COLS = 2


def align_column_shards(shards, n_adv, n_lk, chunk=SET_COLS):
    """The same blocks with their inner boundaries moved (by at most chunk - 1 columns) onto the boundaries of the permutation
    argument's column sets: the permutation runs over [advice | lookup | constants | instance] in sets of `chunk` consecutive
    columns, so an advice cut a must have a % chunk == 0 and a lookup cut l must have (n_adv + l) % chunk == 0.  Then every
    product polynomial's columns lie on one rank — except the one set that spans the advice / lookup junction — and the prover
    rounds shard by the same blocks as the hot path (rounds.ShardMap)."""
    world = len(shards)

    def snap(cuts, offset, n):
        out = [0]
        for c in cuts[1:-1]:
            down = c - (c + offset) % chunk
            up = down + chunk
            c2 = down if (c - down <= up - c or up > n) else up
            out.append(min(n, max(out[-1], c2, 0)))
        out.append(n)
        return out
    a_cuts = snap([0] + [shards[r][0][1] for r in range(world)], 0, n_adv)
    l_cuts = snap([0] + [shards[r][1][1] for r in range(world)], n_adv, n_lk)
    # a lookup cut below the first set that starts inside the lookup columns would split the junction set's columns over three ranks
    first = (-n_adv) % chunk
    l_cuts = [0] + [min(n_lk, max(c, first)) if c > 0 else 0 for c in l_cuts[1:-1]] + [n_lk]
    return [((a_cuts[r], a_cuts[r + 1]), (l_cuts[r], l_cuts[r + 1])) for r in range(world)]


def gather_commitments(dist, local, shards, device):
    """The one real exchange step of the path: all_gather of the 64-byte commitments of every rank's column
    shard (RCCL on GPUs, gloo in the CPU test).  `local`: (my_cols, 8) uint64, [advice block | lookup block];
    `shards`: the per-rank ((a_lo, a_hi), (l_lo, l_hi)) list every rank derived identically (column_shards /
    balanced_column_shards).  Returns (n_adv + n_lk, 8) in the unsharded order [all advice | all lookup]."""
    import torch
    world = len(shards)
    counts = [(a[1] - a[0]) + (l[1] - l[0]) for a, l in shards]
    mx = max(counts)
    mine = torch.zeros((mx, 8), dtype=torch.int64, device=device)
    if len(local):
        mine[: len(local)] = torch.from_numpy(np.ascontiguousarray(local).view(np.int64)).to(device)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    parts = [g.cpu().numpy().view(np.uint64) for g in gathered]
    adv = [p[: a[1] - a[0]] for p, (a, l) in zip(parts, shards)]
    lk = [p[a[1] - a[0]: (a[1] - a[0]) + (l[1] - l[0])] for p, (a, l) in zip(parts, shards)]
    return np.concatenate(adv + lk)


# ---------------------------------------------------------------------------------------------------------------------------
# The alternative partition of SURVEY §8(e): ONE set of columns, the POINTS (rows) split over the ranks.  Only of use when
# there are fewer columns than GPUs (never at the BASELINE configs, where columns are sharded instead); it is what
# BASELINE.json's "all-reduce only for the final bucket reduction" describes.  RCCL has no curve reduction operator: the
# exchange is an all_gather of the per-rank partial commitments (64 B per column and rank) followed by world - 1 group
# additions per column on every rank (vdb_g1_sum) — exact and order independent after the affine normalisation.
def point_shard(rows, rank, world):
    return rows * rank // world, rows * (rank + 1) // world


def point_sharded_partials(srs_shard, cols_dev, n_cols, rows, lo, hi):
    """Commitments of rows [lo, hi) of n_cols device-resident columns (stride `rows`) against an Srs that holds exactly the
    bases of those rows: (n_cols, 8)."""
    lib = srs_shard.L
    desc = np.zeros((n_cols, 3), dtype=np.uint64)                    # vdb_colsrc {src, len, blind}
    base = cols_dev.ptr.value if hasattr(cols_dev, "ptr") else int(cols_dev)
    desc[:, 0] = base + (np.arange(n_cols, dtype=np.uint64) * np.uint64(rows) + np.uint64(lo)) * np.uint64(32)
    desc[:, 1] = hi - lo
    d_src = api.DeviceBuffer(desc.nbytes)
    d_src.upload(desc)
    out = np.zeros((n_cols, 8), dtype=np.uint64)
    try:
        check(lib.vdb_msm_batch_src_dev_begin(srs_shard.h, 1, d_src.ptr, ctypes.c_size_t(n_cols), ctypes.c_size_t(hi - lo), 0, None, None))
        check(lib.vdb_msm_batch_end(api._p(out), ctypes.c_size_t(n_cols)))
    finally:
        d_src.free()
    return out


def combine_partials(parts):
    """(world, n_cols, 8) partial commitments -> (n_cols, 8): the group sum per column (on the device)."""
    parts = np.ascontiguousarray(parts, dtype=np.uint64)
    world, n_cols = parts.shape[0], parts.shape[1]
    out = np.zeros((n_cols, 8), dtype=np.uint64)
    check(api._lib.init().vdb_g1_sum(api._p(parts), ctypes.c_size_t(world), ctypes.c_size_t(n_cols), api._p(out)))
    return out


def allgather_partials(dist, local, device):
    """all_gather of every rank's (n_cols, 8) partial commitments -> (world, n_cols, 8) on the host"""
    import torch
    mine = torch.from_numpy(np.ascontiguousarray(local).view(np.int64)).to(device)
    gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    return np.stack([g.cpu().numpy().view(np.uint64) for g in gathered])


class KmeansHotPath:
    """kmeans::<K, I> over N x D vectors at 2^k rows: witness -> layout -> commit -> NTT, one GPU."""

    def __init__(self, n=256, dim=128, K=4, I=8, k=16, P=48, L=15, metric="euclidean", seed=20260004, tau=None,
                 col_shard=(0, 1), vectors=None, blind_seed=None):
        """`tau`: toxic-waste scalar of the "unsafe" SRS as a canonical integer; None = the scalar the reference's
        `gen_srs(k)` derives from its fixed ChaCha20 seed (srs.gen_srs_tau, src/scaffold/mod.rs:260).
        `vectors`: the f64 rows the circuit assigns first (the `--input` file of the reference's examples,
        src/scaffold/mod.rs:64-77); None = the seeded SIFT-shaped synthetic rows of SURVEY 8(d).
        `blind_seed`: None = the blinding rows of every column are drawn afresh from the operating system's entropy for every
        proof (step), as halo2's create_proof does with its OsRng (reached from src/scaffold/mod.rs:296); an integer fixes
        them — a test hook for comparing commitments across runs, never for real proofs (two proofs that share blinds leak
        the difference of their witnesses)."""
        self.blind_seed = blind_seed
        self._entropy = None
        self.given_vectors = None if vectors is None else np.ascontiguousarray(vectors, dtype=np.float64)
        self.n, self.dim, self.K, self.I, self.k, self.P, self.L = n, dim, K, I, k, P, L
        self.metric = api.METRICS[metric]
        self.metric_name = metric
        self.rows = 1 << k
        self.lib = api.init()
        self.rank, self.world = col_shard
        self.seed = seed
        self.tau = tau
        self.factor_constants = True
        self.shard_witness = True   # generate only the witness cells this rank's columns hold (values are computed everywhere)
        self.balance_shards = True  # equalise estimated time per rank instead of column count
        self.virtual_layout = True  # commit and transform straight from the witness stream (no stream -> column copy)
        self.msm_window_bits = 0    # 0 = the library's default (11 bits for k > 8: most witness scalars are short)
        self.ext_block_cols = None  # columns of extended cosets held at a time; None = all if they fit, else what leaves ext_reserve_bytes free
        self.ext_reserve_bytes = 64 << 30

    # ------------------------------------------------------------------ keygen-like setup (untimed)
    def setup(self, pinning=None):
        """`pinning`: path of a configs/{name}.json written by an earlier keygen (io.write_pinning); when given, its break
        points are used as they are — the Prove arm of the reference (src/scaffold/mod.rs:285-287) — after checking that
        they describe this circuit; otherwise they are derived from the keygen-style run, like the Keygen arm."""
        lib, n, dim, K, I = self.lib, self.n, self.dim, self.K, self.I
        # the MSMs of setup and keygen (constant points, fixed columns) work in a bounded work space: on a still empty card the default
        # (half of the free HBM) would map up to 96 GiB for a two-second MSM, and mapping fresh HBM costs ~30 ms / GiB; step() lifts it
        api.msm_scratch_cap(api.KEYGEN_SCRATCH_CAP)
        self._cap_lifted = False
        vec, self.seed = self._input_vectors() if self.given_vectors is None else (self.given_vectors, self.seed)
        assert vec.shape == (self.n_input_rows(), self.dim), "input rows do not match the circuit's shape"
        self.vectors_f64 = vec
        self.qvec = api.quantize(vec, self.P)
        self.n_in, n_gadget_cells, self.n_lookup = self._circuit_size()
        self.n_cells = self.n_in + n_gadget_cells
        B = 32
        self.d_vec = api.DeviceBuffer(self.qvec.nbytes)
        self.d_vec.upload(self.qvec)
        self.d_stream = api.DeviceBuffer(self.n_cells * B)
        self.d_lookup = api.DeviceBuffer(max(self.n_lookup, 1) * B)
        self._alloc_outputs()
        # keygen-style run: record gate starts, derive break points (the reference pins them in configs/*.json)
        d_sel = api.DeviceBuffer(self.n_cells)
        check(lib.vdb_memset_dev(d_sel.ptr, 0, ctypes.c_size_t(self.n_cells)))
        self._witness(sel=d_sel)
        nbp = ctypes.c_uint64()
        check(lib.vdb_layout_plan_dev(d_sel.ptr, ctypes.c_uint64(self.n_cells), self.k, MINIMUM_ROWS, None, ctypes.c_uint64(0), ctypes.byref(nbp)))
        self.bp = np.zeros(max(nbp.value, 1), dtype=np.uint64)
        check(lib.vdb_layout_plan_dev(d_sel.ptr, ctypes.c_uint64(self.n_cells), self.k, MINIMUM_ROWS, api._p(self.bp), ctypes.c_uint64(self.bp.size),
                                      ctypes.byref(nbp)))
        self.bp = self.bp[:nbp.value]
        if pinning is not None:
            from .io import read_pinning
            params, bp = read_pinning(pinning)
            if params["degree"] != self.k or params["lookup_bits"] != self.L or not np.array_equal(bp, self.bp):
                raise ValueError("pinning file does not describe this circuit (degree, lookup_bits or break points differ)")
            self.bp = bp
        self.n_adv_cols = len(self.bp) + 1
        max_rows = self.rows - MINIMUM_ROWS
        self.n_lk_cols = math.ceil(self.n_lookup / max_rows)
        self.n_cols = self.n_adv_cols + self.n_lk_cols
        from_ints = lambda vals: np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
        R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
        if self.tau is None:
            from .srs import gen_srs_tau
            self.tau = gen_srs_tau()
        tau = from_ints([self.tau * (1 << 256) % R])[0]
        g, gl = api.srs_setup_unsafe(self.k, tau)
        self.g_lagrange = gl
        self.g_monomial = g
        self.srs = api.Srs(self.k, None, gl, window_bits=self.msm_window_bits)
        # column sharding over ranks: a block of the advice columns and a block of the lookup columns each
        self.shards = column_shards(self.n_adv_cols, self.n_lk_cols, self.world)
        if self.world > 1 and self.balance_shards:
            # keygen-time statistics of the layout just produced (every rank computes the same numbers)
            n_el = self.n_adv_cols * self.rows
            d_tmpm = api.DeviceBuffer(n_el)
            check(lib.vdb_layout_const_mask_dev(d_sel.ptr, ctypes.c_uint64(self.n_cells), api._p(self.bp), ctypes.c_uint64(len(self.bp)), self.k, d_tmpm.ptr))
            d_tmpc = api.DeviceBuffer(max(self.n_adv_cols, self.n_lk_cols, 1) * self.rows * B)
            check(lib.vdb_layout_columns_dev(self.d_stream.ptr, ctypes.c_uint64(self.n_cells), api._p(self.bp), ctypes.c_uint64(len(self.bp)), self.k,
                                             d_tmpc.ptr, None, 0))
            var_adv = np.zeros(self.n_adv_cols, dtype=np.uint64)
            check(lib.vdb_msm_count_entries_dev(self.srs.h, d_tmpc.ptr, ctypes.c_size_t(self.n_adv_cols), ctypes.c_size_t(self.rows), d_tmpm.ptr,
                                                api._p(var_adv)))
            var_lk = np.zeros(max(self.n_lk_cols, 1), dtype=np.uint64)
            if self.n_lk_cols:
                check(lib.vdb_layout_lookup_dev(self.d_lookup.ptr, ctypes.c_uint64(self.n_lookup), self.k, MINIMUM_ROWS, d_tmpc.ptr,
                                                ctypes.c_uint64(self.n_lk_cols), None, 0))
                check(lib.vdb_msm_count_entries_dev(self.srs.h, d_tmpc.ptr, ctypes.c_size_t(self.n_lk_cols), ctypes.c_size_t(self.rows), None,
                                                    api._p(var_lk)))
            d_tmpm.free()
            d_tmpc.free()
            self.msm_entries = (var_adv, var_lk[: self.n_lk_cols])
            self.shards = balanced_column_shards(var_adv, var_lk[: self.n_lk_cols], self.world)
        if self.world > 1:
            self.shards = align_column_shards(self.shards, self.n_adv_cols, self.n_lk_cols, set_cols(self.n_lk_cols))
        (self.a_lo, self.a_hi), (self.l_lo, self.l_hi) = self.shards[self.rank]
        self.my_adv, self.my_lk = self.a_hi - self.a_lo, self.l_hi - self.l_lo
        self.my_cols = self.my_adv + self.my_lk
        # the stretch of the flat stream / lookup stream those columns hold (column c = stream[starts[c] : starts[c] + bp[c] + 1])
        starts = np.concatenate([[0], np.cumsum(self.bp, dtype=np.uint64)]).astype(np.uint64)
        if self.my_adv:
            s_hi = int(starts[self.a_hi - 1]) + int(self.bp[self.a_hi - 1]) + 1 if self.a_hi - 1 < len(self.bp) else self.n_cells
            self.win_adv = (int(starts[self.a_lo]), s_hi)
        else:
            self.win_adv = (0, 0)
        self.win_lk = (self.l_lo * max_rows, min(self.l_hi * max_rows, self.n_lookup)) if self.my_lk else (0, 0)
        # N_BLIND blinding scalars per column, uniform in Fr: 64 bytes of entropy each, reduced on the device
        self.d_blind = api.DeviceBuffer(self.n_cols * N_BLIND * B)
        self.d_wide = api.DeviceBuffer(self.n_cols * N_BLIND * 64)
        self.refresh_blinds()
        self.d_cols = api.DeviceBuffer(max(self.my_cols, 1) * self.rows * B)
        # column descriptors of my [advice block | lookup block]: where each column lies in the streams (data independent)
        self.d_src = api.DeviceBuffer(max(self.my_cols, 1) * 24)
        if self.my_adv:
            check(lib.vdb_colsrc_build_dev(self.d_stream.ptr, ctypes.c_uint64(self.n_cells), api._p(self.bp), ctypes.c_uint64(len(self.bp)), self.k,
                                           ctypes.c_uint64(self.a_lo), ctypes.c_uint64(self.a_hi), self.d_blind.ptr, N_BLIND, self.d_src.ptr))
        if self.my_lk:
            check(lib.vdb_colsrc_build_lookup_dev(self.d_lookup.ptr, ctypes.c_uint64(self.n_lookup), self.k, MINIMUM_ROWS, ctypes.c_uint64(self.l_lo),
                                                  ctypes.c_uint64(self.l_hi), self.d_blind.at(self.n_adv_cols * N_BLIND * B), N_BLIND,
                                                  self.d_src.at(self.my_adv * 24)))
        # keygen-time factoring of the constant cells: column-layout mask of the QuantumCell::Constant cells and the
        # per-column MSM of exactly those cells (data independent, so computed once like the rest of the proving key)
        # (lookup columns hold no constants: their mask stays zero and their constant point is the identity).
        # Only this rank's advice columns, laid out in the column buffer the step overwrites anyway: no second buffer of the
        # columns' size (33.7 GiB at C4') is allocated and handed back.
        n_el = self.n_adv_cols * self.rows
        d_fmask = api.DeviceBuffer(n_el)                    # one byte per cell of ALL advice columns (1.1 GiB at C4')
        check(lib.vdb_layout_const_mask_dev(d_sel.ptr, ctypes.c_uint64(self.n_cells), api._p(self.bp), ctypes.c_uint64(len(self.bp)), self.k, d_fmask.ptr))
        d_sel.free()
        self.d_mask = api.DeviceBuffer(max(self.my_cols, 1) * self.rows)
        check(lib.vdb_memset_dev(self.d_mask.ptr, 0, ctypes.c_size_t(max(self.my_cols, 1) * self.rows)))
        const_mine = np.zeros((self.my_cols, 8), dtype=np.uint64)
        if self.my_adv:
            check(lib.vdb_memcpy_d2d(self.d_mask.ptr, d_fmask.at(self.a_lo * self.rows), ctypes.c_size_t(self.my_adv * self.rows)))
            check(lib.vdb_layout_columns_range_dev(self.d_stream.ptr, ctypes.c_uint64(self.n_cells), api._p(self.bp), ctypes.c_uint64(len(self.bp)), self.k,
                                                   ctypes.c_uint64(self.a_lo), ctypes.c_uint64(self.a_hi), self.d_cols.ptr, None, 0))
            check(lib.vdb_mask_select_dev(self.d_cols.ptr, self.d_mask.ptr, ctypes.c_uint64(self.my_adv * self.rows), 1, self.d_cols.ptr))
            pts = np.zeros((self.my_adv, 8), dtype=np.uint64)
            check(lib.vdb_msm_batch_dev(self.srs.h, 1, self.d_cols.ptr, ctypes.c_size_t(self.my_adv), ctypes.c_size_t(self.rows), api._p(pts)))
            const_mine[: self.my_adv] = pts
        self.const_points = const_mine                      # [my advice | my lookup] order, like d_cols
        self.d_cpts = api.DeviceBuffer(max(const_mine.nbytes, 64))
        if const_mine.nbytes:
            self.d_cpts.upload(np.ascontiguousarray(const_mine))
        self.const_cell_fraction = float(d_fmask.download((n_el,), dtype=np.uint8).mean()) if n_el <= (1 << 28) else None
        d_fmask.free()
        api.sync()
        # The extended cosets, last: all of them when they fit beside everything above and leave the MSM its work space (C4:
        # 68 GB), otherwise the largest block of columns that does — the cosets are then produced block after block into the
        # same buffer (a circuit larger than HBM streams through: C4' cosine, 20.3 k columns = 170 GB of cosets; C5's Merkle).
        # Two columns more than this rank holds: the prover rounds keep the cosets of the constants' fixed column and of the
        # instance column behind the advice cosets, so that the permutation argument reads its columns from one contiguous
        # block (rounds.py).
        per_col = self.rows * 4 * B
        want = max(self.my_cols, 1) + 2
        if self.ext_block_cols is None:
            free, _ = api.mem_info()
            free += api.scratch_held()       # (the bounded work space stays allocated: it is part of what ext_reserve_bytes leaves the MSM)
            fit = (free - self.ext_reserve_bytes) // per_col
            self.ext_cols = want if fit >= want else int(max(min(want, 256), fit // 256 * 256))
        else:
            self.ext_cols = min(want, int(self.ext_block_cols))
        self.d_ext = api.DeviceBuffer(self.ext_cols * per_col)
        return self

    # ------------------------------------------------------------------ blinding
    def _draw_entropy(self, seed):
        import os
        nbytes = self.n_cols * N_BLIND * 64
        return os.urandom(nbytes) if seed is None else np.random.default_rng(seed).bytes(nbytes)

    def refresh_blinds(self, seed=None):
        """New blinding rows for every column (halo2: Blind(Scalar::random(OsRng)) per committed polynomial).  The bytes for
        the next proof are drawn while the GPU is busy with this one (step()), so a refresh costs one small upload."""
        seed = self.blind_seed if seed is None else seed
        raw = self._entropy if (seed is None and self._entropy is not None) else self._draw_entropy(seed)
        self._entropy = None
        self.d_wide.upload(np.frombuffer(raw, dtype=np.uint8))
        check(self.lib.vdb_fr_from_wide_dev(self.d_wide.ptr, ctypes.c_size_t(self.n_cols * N_BLIND), self.d_blind.ptr))

    # ------------------------------------------------------------------ what is specific to the k-means circuit
    def _input_vectors(self):
        """(f64 rows that ctx.assign_witnesses puts at the head of the stream, seed actually used)"""
        return sift_like_vectors(self.seed, self.n, self.dim, self.K)

    def n_input_rows(self):
        return self.n

    def _circuit_size(self):
        """(cells of ctx.assign_witnesses(quantize_vector(v)) for every vector, cells the gadget emits, lookup cells)"""
        cells, lk = ctypes.c_uint64(), ctypes.c_uint64()
        check(self.lib.vdb_wit_kmeans_size(self.metric, self.P, self.L, self.n, self.dim, self.K, self.I, 0, ctypes.byref(cells), ctypes.byref(lk)))
        return self.n * self.dim, cells.value, lk.value

    def _alloc_outputs(self):
        self.d_cent = api.DeviceBuffer(self.K * self.dim * 32)
        self.d_ind = api.DeviceBuffer(self.n * self.K * 32)

    def write_pinning(self, path):
        """configs/{name}.json of the Keygen arm (src/scaffold/mod.rs:272)."""
        from .io import write_pinning
        write_pinning(path, self.k, self.bp, self.n_lk_cols, self.L)

    def set_vectors(self, vectors_f64):
        """Prove for a different database of the same shape (the keygen-time data above stays untouched)."""
        self.vectors_f64 = np.asarray(vectors_f64, dtype=np.float64)
        self.qvec = api.quantize(self.vectors_f64, self.P)
        self.d_vec.upload(self.qvec)

    def _layout(self, dest=None):
        """My block of advice columns followed by my block of lookup columns, compact in d_cols (or `dest`)."""
        lib, B = self.lib, 32
        d_cols = self.d_cols if dest is None else dest
        if self.my_adv:
            check(lib.vdb_layout_columns_range_dev(self.d_stream.ptr, ctypes.c_uint64(self.n_cells), api._p(self.bp), ctypes.c_uint64(len(self.bp)), self.k,
                                                   ctypes.c_uint64(self.a_lo), ctypes.c_uint64(self.a_hi), d_cols.ptr, self.d_blind.ptr, N_BLIND))
        if self.my_lk:
            check(lib.vdb_layout_lookup_range_dev(self.d_lookup.ptr, ctypes.c_uint64(self.n_lookup), self.k, MINIMUM_ROWS, ctypes.c_uint64(self.l_lo),
                                                  ctypes.c_uint64(self.l_hi), d_cols.at(self.my_adv * self.rows * B),
                                                  self.d_blind.at(self.n_adv_cols * N_BLIND * B), N_BLIND))

    def _witness(self, sel=None):
        lib = self.lib
        # [assign_witnesses(vectors)] [kmeans cells]
        if self.n_in:
            check(lib.vdb_memcpy_d2d(self.d_stream.ptr, self.d_vec.ptr, ctypes.c_size_t(self.n_in * 32)))
        windowed = sel is None and self.shard_witness and self.world > 1
        if windowed:
            # window in the coordinates of the pointers handed to the call (the kmeans cells start n_in cells into the stream)
            lo, hi = (max(0, x - self.n_in) for x in self.win_adv)
            check(lib.vdb_wit_set_window(ctypes.c_uint64(lo), ctypes.c_uint64(hi), ctypes.c_uint64(self.win_lk[0]), ctypes.c_uint64(self.win_lk[1])))
        try:
            check(lib.vdb_wit_kmeans_dev(self.metric, self.P, self.L, self.d_vec.ptr, self.n, self.dim, self.K, self.I, 0, self.d_stream.at(self.n_in * 32),
                                         self.d_lookup.ptr, ctypes.c_void_p(sel.ptr.value + self.n_in) if sel is not None else None, self.d_cent.ptr,
                                         self.d_ind.ptr))
        finally:
            if windowed:
                check(lib.vdb_wit_set_window(ctypes.c_uint64(0), ctypes.c_uint64(2**64 - 1), ctypes.c_uint64(0), ctypes.c_uint64(2**64 - 1)))

    # ------------------------------------------------------------------ one pass of the hot path
    def step(self, timings=None, blind_seed=None, with_ext=True, sync=True, after_witness=None):
        """`with_ext`: False leaves coeff_to_extended out (a caller that streams the cosets itself, block by block: rounds.py on
        circuits whose cosets do not fit HBM).  `sync`: False returns as soon as the commitments are on the host, the transforms
        still running on the library's stream (the caller absorbs the commitments into its transcript meanwhile; everything it
        queues next is ordered behind them).  `after_witness`: called once the witness kernels are queued and before the
        commitments are (the prover rounds read the public cells out of the stream there)."""
        lib, B = self.lib, 32
        if not getattr(self, "_cap_lifted", True):      # the prover's MSM takes the work space its default policy gives it
            api.msm_scratch_cap(0)
            self._cap_lifted = True
        self.refresh_blinds(blind_seed)

        def stage(name, fn):
            if timings is not None:
                api.timer_start()
            fn()
            if timings is not None:
                timings[name] = timings.get(name, 0.0) + api.timer_stop()

        stage("witness", self._witness)
        if after_witness is not None:
            after_witness()

        virt = self.virtual_layout and self.k > 10
        if not virt:
            stage("layout", self._layout)
        my = self.d_cols.ptr
        self.commitments = np.zeros((self.my_cols, 8), dtype=np.uint64)

        def commit():
            # the MSM is queued without waiting; the bucket folding of its last batch runs on a second stream beside the NTTs
            mask = self.d_mask.ptr if self.factor_constants else None
            cpts = self.d_cpts.ptr if self.factor_constants else None
            if virt:
                check(lib.vdb_msm_batch_src_dev_begin(self.srs.h, 1, self.d_src.ptr, ctypes.c_size_t(self.my_cols), ctypes.c_size_t(self.rows), N_BLIND, mask, cpts))
            else:
                check(lib.vdb_msm_batch_masked_dev_begin(self.srs.h, 1, my, ctypes.c_size_t(self.my_cols), ctypes.c_size_t(self.rows), mask, cpts))

        def ntt():
            if virt:
                check(lib.vdb_lagrange_to_coeff_src_dev(self.d_src.ptr, my, ctypes.c_size_t(self.my_cols), self.k, N_BLIND))
            else:
                check(lib.vdb_lagrange_to_coeff_dev(my, ctypes.c_size_t(self.my_cols), self.k))
            # cosets of all columns when the buffer holds them, else block after block into the same buffer
            blk = min(self.ext_cols, max(self.my_cols, 1))
            for c0 in range(0, self.my_cols if with_ext else 0, blk):
                nb = min(blk, self.my_cols - c0)
                check(lib.vdb_coeff_to_extended_dev(self.d_cols.at(c0 * self.rows * B), self.d_ext.ptr, ctypes.c_size_t(nb), self.k, 2))
            if self.blind_seed is None and blind_seed is None:
                self._entropy = self._draw_entropy(None)      # the next proof's blinds, drawn while the GPU works on this one
            check(lib.vdb_msm_batch_end(api._p(self.commitments), ctypes.c_size_t(self.my_cols)))

        stage("commit_msm", commit)
        try:
            stage("ntt", ntt)
        except Exception:
            # a deferred MSM must always be collected, or every later MSM is refused ("deferred batch has not been collected")
            lib.vdb_msm_batch_end(None, ctypes.c_size_t(0))
            raise
        if sync:
            api.sync()
        return self.commitments

    # ------------------------------------------------------------------ the Mock stage (src/scaffold/mod.rs:263-266)
    def keygen_flags(self):
        """flag byte per advice cell from a keygen-style run (bit 0 gate start, bit 1 constant cell, bit 2 lookup source), on the device"""
        d_flags = api.DeviceBuffer(self.n_cells)
        check(self.lib.vdb_memset_dev(d_flags.ptr, 0, ctypes.c_size_t(self.n_cells)))
        self._witness(sel=d_flags)     # the flags are data independent; the cells written are those of the current input
        return d_flags

    def mock_check(self, copy_of=None, lookup_src=None, const_stream=None):
        """MockProver::run(..).assert_satisfied() of the reference's Mock arm on the witness as it lies in HBM after a
        step() / relayout(): every gate row, every lookup cell against the range table and — when the maps are given (device
        buffers or int64 arrays) — every copy constraint and every constant.  Returns api.MockReport (rank-unsharded runs)."""
        assert self.world == 1, "the mock check walks the whole witness"
        d_flags = self.keygen_flags()
        self._witness()           # the witness under test (the flag run above wrote the same cells)
        own = []

        def dev(a, dtype):
            if a is None or isinstance(a, (api.DeviceBuffer,)) or hasattr(a, "ptr"):
                return a
            a = np.ascontiguousarray(a, dtype=dtype)
            b = api.DeviceBuffer(max(a.nbytes, 32))
            b.upload(a)
            own.append(b)
            return b
        try:
            c, l, k = dev(copy_of, np.int64), dev(lookup_src, np.int64), dev(const_stream, np.uint64)
            return api.mock_check_dev(self.d_stream.ptr, self.n_cells, d_flags.ptr, self.d_lookup.ptr, self.n_lookup, self.L,
                                      None if c is None else c.ptr, None if l is None else l.ptr, None if k is None else k.ptr)
        finally:
            d_flags.free()
            for b in own:
                b.free()

    def local_index(self, col):
        """Position in this rank's compact column buffer of global column `col` ([all advice | all lookup] numbering)."""
        if col < self.n_adv_cols:
            assert self.a_lo <= col < self.a_hi, "column not held by this rank"
            return col - self.a_lo
        c = col - self.n_adv_cols
        assert self.l_lo <= c < self.l_hi, "column not held by this rank"
        return self.my_adv + c - self.l_lo

    def global_columns(self):
        """Global numbers of the columns this rank holds, in buffer order."""
        return list(range(self.a_lo, self.a_hi)) + [self.n_adv_cols + c for c in range(self.l_lo, self.l_hi)]

    def download_columns(self, col_indices):
        """Lagrange-basis columns as laid out (call right after `layout`, i.e. use relayout()); global column numbers."""
        out = np.zeros((len(col_indices), self.rows, 4), dtype=np.uint64)
        for j, c in enumerate(col_indices):
            out[j] = self.d_cols.download((self.rows, 4), offset=self.local_index(c) * self.rows * 32)
        return out

    def relayout(self):
        """Re-run witness + layout only (columns are overwritten in place by the NTT stage)."""
        self._witness()
        self._layout()
        api.sync()

    def public_values_dev(self):
        """(device pointer, count) of the values the reference's example of this circuit makes public, in make_public order
        (examples/kmeans.rs:51-56: every centroid, word by word).  Every rank computes them (value-only walk), whichever rank's
        columns hold the cells."""
        return self.d_cent.ptr, self.K * self.dim

    def results(self):
        cent = self.d_cent.download((self.K, self.dim, 4))
        ind = self.d_ind.download((self.n, self.K, 4))
        return cent, ind

    def free(self):
        for name in ("d_vec", "d_stream", "d_lookup", "d_cent", "d_ind", "d_blind", "d_wide", "d_cols", "d_ext", "d_mask", "d_cpts", "d_src"):
            b = getattr(self, name, None)
            if b is not None:
                b.free()
        if getattr(self, "srs", None) is not None:
            self.srs.free()


class MerkleHotPath(KmeansHotPath):
    """merkle_commitment over N x D vectors (src/gadget/vectordb.rs:165-223; examples/merkle.rs) through the same hot
    path: Poseidon trace on the GPU -> commit -> NTT.  No lookup cells; every column is dense (Poseidon states), so equal
    column counts per rank are balanced already; a rank that holds a block of columns traces only the permutations whose
    cells fall into it (the sponge states and the tree's digests are computed by every rank, value only)."""

    def __init__(self, n=1024, dim=128, k=15, P=32, seed=20260003, tau=None, col_shard=(0, 1), vectors=None, blind_seed=None):
        super().__init__(n=n, dim=dim, K=1, I=1, k=k, P=P, L=8, seed=seed, tau=tau, col_shard=col_shard, vectors=vectors, blind_seed=blind_seed)
        self.balance_shards = False
        self.msm_window_bits = 14   # every scalar is a full-width Poseidon state: 19 windows instead of 24

    def _circuit_size(self):
        cells = ctypes.c_uint64()
        check(self.lib.vdb_wit_merkle_size(self.n, self.dim, 0, ctypes.byref(cells)))
        return self.n * self.dim, cells.value, 0      # the vectors are assigned first (tests/vectordb/mod.rs:253-262 chip_merkle)

    def _alloc_outputs(self):
        self.d_root = api.DeviceBuffer(32)

    def _witness(self, sel=None):
        lib = self.lib
        check(lib.vdb_memcpy_d2d(self.d_stream.ptr, self.d_vec.ptr, ctypes.c_size_t(self.n_in * 32)))
        windowed = sel is None and self.shard_witness and self.world > 1
        if windowed:   # only the permutations whose cells fall into this rank's columns are traced; the digests are computed everywhere
            lo, hi = (max(0, x - self.n_in) for x in self.win_adv)       # in the coordinates of the pointer handed to the call
            check(lib.vdb_wit_set_window(ctypes.c_uint64(lo), ctypes.c_uint64(hi), ctypes.c_uint64(0), ctypes.c_uint64(0)))
        try:
            check(lib.vdb_wit_merkle_dev(self.d_vec.ptr, self.n, self.dim, 0, self.d_stream.at(self.n_in * 32),
                                         ctypes.c_void_p(sel.ptr.value + self.n_in) if sel is not None else None, self.d_root.ptr))
        finally:
            if windowed:
                check(lib.vdb_wit_set_window(ctypes.c_uint64(0), ctypes.c_uint64(2**64 - 1), ctypes.c_uint64(0), ctypes.c_uint64(2**64 - 1)))

    def public_values_dev(self):
        return self.d_root.ptr, 1                  # examples/merkle.rs:47

    def results(self):
        return self.d_root.download((4,))

    def free(self):
        super().free()
        if getattr(self, "d_root", None) is not None:
            self.d_root.free()
            self.d_root = None


class NearestHotPath(KmeansHotPath):
    """nearest_vector(query, vectors) (src/gadget/vectordb.rs:122-163; tests/vectordb/mod.rs:220-247 assigns the query, then the
    vectors) through the same hot path.  Sharded: every rank computes the N distances' values and the short minimum
    chain (value-only walk), and stores the cells of its own block of columns only — the distance blocks, N-way parallel and nearly
    all of the cells, are skipped outside the rank's window."""

    def __init__(self, n=64, dim=128, k=14, P=48, L=13, metric="euclidean", seed=20260002, tau=None, col_shard=(0, 1), vectors=None, blind_seed=None):
        """`vectors`: (n + 1, dim) f64 rows, the query first"""
        super().__init__(n=n, dim=dim, K=1, I=1, k=k, P=P, L=L, metric=metric, seed=seed, tau=tau, col_shard=col_shard, vectors=vectors,
                         blind_seed=blind_seed)

    def n_input_rows(self):
        return self.n + 1

    def _input_vectors(self):
        vec, seed = sift_like_vectors(self.seed, self.n, self.dim)
        query, _ = sift_like_vectors(seed + 1000, 1, self.dim)
        return np.concatenate([query, vec]), seed

    def _circuit_size(self):
        cells, lk = ctypes.c_uint64(), ctypes.c_uint64()
        check(self.lib.vdb_wit_nearest_size(self.metric, self.P, self.L, self.n, self.dim, ctypes.byref(cells), ctypes.byref(lk)))
        return (self.n + 1) * self.dim, cells.value, lk.value

    def _alloc_outputs(self):
        self.d_ind = api.DeviceBuffer(self.n * 32)
        self.d_res = api.DeviceBuffer(self.dim * 32)

    def _witness(self, sel=None):
        lib = self.lib
        check(lib.vdb_memcpy_d2d(self.d_stream.ptr, self.d_vec.ptr, ctypes.c_size_t(self.n_in * 32)))
        windowed = sel is None and self.shard_witness and self.world > 1
        if windowed:   # in the coordinates of the pointers handed to the call (the gadget's cells start n_in cells into the stream)
            lo, hi = (max(0, x - self.n_in) for x in self.win_adv)
            check(lib.vdb_wit_set_window(ctypes.c_uint64(lo), ctypes.c_uint64(hi), ctypes.c_uint64(self.win_lk[0]), ctypes.c_uint64(self.win_lk[1])))
        try:
            check(lib.vdb_wit_nearest_dev(self.metric, self.P, self.L, self.d_vec.ptr, self.d_vec.at(self.dim * 32), self.n, self.dim, self.d_stream.at(self.n_in * 32),
                                          self.d_lookup.ptr, ctypes.c_void_p(sel.ptr.value + self.n_in) if sel is not None else None, self.d_ind.ptr,
                                          self.d_res.ptr))
        finally:
            if windowed:
                check(lib.vdb_wit_set_window(ctypes.c_uint64(0), ctypes.c_uint64(2**64 - 1), ctypes.c_uint64(0), ctypes.c_uint64(2**64 - 1)))

    def public_values_dev(self):
        return self.d_res.ptr, self.dim            # examples/query.rs:58: the nearest vector

    def results(self):
        return self.d_ind.download((self.n, 4)), self.d_res.download((self.dim, 4))

    def free(self):
        super().free()
        if getattr(self, "d_res", None) is not None:
            self.d_res.free()
            self.d_res = None


class QueryHotPath(NearestHotPath):
    """The reference's `query` circuit — "exhaustively find the similar vector & commit to the database" (examples/query.rs:32-73;
    tests/vectordb/mod.rs:220-247 chip_nearest_vector): nearest_vector(query, database) and merkle_commitment(database) in ONE
    circuit over the same assigned vectors, the result vector and the Merkle root public.  Stream: [query | vectors | nearest_vector's
    cells | merkle_commitment's cells]; the lookup cells are nearest_vector's.  (PoseidonChip::new's three load_constant cells are
    not emitted, as in MerkleHotPath: the sponge's initial state is pinned as constants of the circuit.)"""

    def _circuit_size(self):
        n_in, nv_cells, lk = super()._circuit_size()
        mk = ctypes.c_uint64()
        check(self.lib.vdb_wit_merkle_size(self.n, self.dim, 0, ctypes.byref(mk)))
        self.nearest_cells, self.merkle_cells = nv_cells, mk.value
        return n_in, nv_cells + mk.value, lk

    def _alloc_outputs(self):
        super()._alloc_outputs()
        self.d_root = api.DeviceBuffer(32)
        self.d_pub = api.DeviceBuffer((self.dim + 1) * 32)      # [result vector | root]: the public statement, in make_public order

    def _witness(self, sel=None):
        lib = self.lib
        super()._witness(sel)                                       # inputs + nearest_vector (windowed like NearestHotPath)
        off = self.n_in + self.nearest_cells
        windowed = sel is None and self.shard_witness and self.world > 1
        if windowed:                                                 # in the coordinates of the Merkle trace's first cell
            lo, hi = (max(0, x - off) for x in self.win_adv)
            check(lib.vdb_wit_set_window(ctypes.c_uint64(lo), ctypes.c_uint64(hi), ctypes.c_uint64(0), ctypes.c_uint64(0)))
        try:
            check(lib.vdb_wit_merkle_dev(self.d_vec.at(self.dim * 32), self.n, self.dim, 0, self.d_stream.at(off * 32),
                                         ctypes.c_void_p(sel.ptr.value + off) if sel is not None else None, self.d_root.ptr))
        finally:
            if windowed:
                check(lib.vdb_wit_set_window(ctypes.c_uint64(0), ctypes.c_uint64(2**64 - 1), ctypes.c_uint64(0), ctypes.c_uint64(2**64 - 1)))
        check(lib.vdb_memcpy_d2d(self.d_pub.ptr, self.d_res.ptr, ctypes.c_size_t(self.dim * 32)))
        check(lib.vdb_memcpy_d2d(self.d_pub.at(self.dim * 32), self.d_root.ptr, ctypes.c_size_t(32)))

    def public_values_dev(self):
        return self.d_pub.ptr, self.dim + 1          # examples/query.rs:58 make_public.extend(result), :69 make_public.push(root)

    def results(self):
        ind, res = super().results()
        return ind, res, self.d_root.download((4,))

    def free(self):
        super().free()
        for name in ("d_root", "d_pub"):
            if getattr(self, name, None) is not None:
                getattr(self, name).free()
                setattr(self, name, None)


class DistancesHotPath(KmeansHotPath):
    """The reference's two-vector circuits through the same hot path: examples/distances.rs:29-59 (assign a, assign b, then
    euclidean, manhattan, cosine and hamming distance of the same two vectors, each made public) and examples/euclid.rs:26-46 (ten Euclidean distances of one pair, nothing public:
    `metrics=("euclidean",) * 10, public=False`).  BASELINE configs[0] is this circuit with one Euclidean distance of two 4-dim
    vectors at k = 13, LOOKUP_BITS = 12.  Stream: [a | b | the cells of each distance in turn]; every rank emits every cell (a few
    columns: nothing to shard the witness by)."""

    def __init__(self, dim=4, metrics=("euclidean", "manhattan", "cosine", "hamming"), k=13, P=48, L=12, seed=20260001, tau=None, col_shard=(0, 1), vectors=None,
                 blind_seed=None, public=True):
        """`vectors`: (2, dim) f64 rows, a then b"""
        super().__init__(n=2, dim=dim, K=1, I=1, k=k, P=P, L=L, metric=metrics[0], seed=seed, tau=tau, col_shard=col_shard, vectors=vectors,
                         blind_seed=blind_seed)
        self.metrics = tuple(metrics)
        for m in self.metrics:
            if m not in api.METRICS:
                raise ValueError("unknown distance: " + str(m))
        self.public = bool(public)
        self.shard_witness = False

    def _input_vectors(self):
        return sift_like_vectors(self.seed, 2, self.dim)

    def _circuit_size(self):
        self.parts = []
        cells_total = lk_total = 0
        for m in self.metrics:
            cells, lk = ctypes.c_uint64(), ctypes.c_uint64()
            check(self.lib.vdb_wit_distance_size(api.METRICS[m], self.P, self.L, 1, self.dim, ctypes.byref(cells), ctypes.byref(lk)))
            self.parts.append((api.METRICS[m], cells_total, lk_total))
            cells_total, lk_total = cells_total + cells.value, lk_total + lk.value
        return 2 * self.dim, cells_total, lk_total

    def _alloc_outputs(self):
        self.d_res = api.DeviceBuffer(len(self.metrics) * 32)

    def _witness(self, sel=None):
        lib = self.lib
        check(lib.vdb_memcpy_d2d(self.d_stream.ptr, self.d_vec.ptr, ctypes.c_size_t(self.n_in * 32)))
        for i, (metric, off, lk_off) in enumerate(self.parts):
            at = self.n_in + off
            check(lib.vdb_wit_distance_dev(metric, self.P, self.L, self.d_vec.ptr, self.d_vec.at(self.dim * 32), 1, self.dim, self.d_stream.at(at * 32),
                                           self.d_lookup.at(lk_off * 32), ctypes.c_void_p(sel.ptr.value + at) if sel is not None else None,
                                           self.d_res.at(i * 32)))

    def public_values_dev(self):
        return self.d_res.ptr, len(self.metrics) if self.public else 0      # examples/distances.rs:44-59: make_public.push(dist) after each

    def results(self):
        return self.d_res.download((len(self.metrics), 4))

    def free(self):
        super().free()
        if getattr(self, "d_res", None) is not None:
            self.d_res.free()
            self.d_res = None



class FixedPointHotPath(DistancesHotPath):
    """examples/fixed_point.rs:38-112 through the hot path: FixedPointChip<32> on ONE value — x = ctx.load_witness(quantization(x)), then
    qexp2(x), qlog2(x) when x > 0, qsin(x), with x and every result public.  Stream: [x | the cells of each call in turn]
    (vdb_wit_fp_op_dev, one instance each).  `ops`: other FixedPointInstructions names instead of the example's."""

    def __init__(self, x=1.128, ops=None, k=13, P=32, L=12, tau=None, blind_seed=None):
        self.x = float(x)
        self.ops = tuple(ops) if ops is not None else ("qexp2",) + (("qlog2",) if self.x > 0.0 else ()) + ("qsin",)
        for name in self.ops:
            if name not in api.FP_OPS:
                raise ValueError("unknown FixedPointChip operation: " + str(name))
        KmeansHotPath.__init__(self, n=1, dim=1, K=1, I=1, k=k, P=P, L=L, metric="euclidean", tau=tau, vectors=np.array([[self.x]]), blind_seed=blind_seed)
        self.metrics = ()
        self.public = True
        self.shard_witness = False

    def _input_vectors(self):
        return np.array([[self.x]], dtype=np.float64), None

    def _circuit_size(self):
        self.parts = []
        cells_total = lk_total = 0
        for name in self.ops:
            cells, lk = ctypes.c_uint64(), ctypes.c_uint64()
            check(self.lib.vdb_wit_fp_op_size(api.FP_OPS[name], self.P, self.L, 1, ctypes.byref(cells), ctypes.byref(lk)))
            self.parts.append((api.FP_OPS[name], cells_total, lk_total))
            cells_total, lk_total = cells_total + cells.value, lk_total + lk.value
        return 1, cells_total, lk_total

    def _alloc_outputs(self):
        self.d_res = api.DeviceBuffer((1 + len(self.ops)) * 32)        # [x | results]: the public statement in make_public order

    def _witness(self, sel=None):
        lib = self.lib
        check(lib.vdb_memcpy_d2d(self.d_stream.ptr, self.d_vec.ptr, ctypes.c_size_t(32)))
        check(lib.vdb_memcpy_d2d(self.d_res.ptr, self.d_vec.ptr, ctypes.c_size_t(32)))
        for i, (op, off, lk_off) in enumerate(self.parts):
            at = 1 + off
            check(lib.vdb_wit_fp_op_dev(op, self.P, self.L, self.d_vec.ptr, None, 1, self.d_stream.at(at * 32), self.d_lookup.at(lk_off * 32),
                                        ctypes.c_void_p(sel.ptr.value + at) if sel is not None else None, self.d_res.at((1 + i) * 32)))

    def public_values_dev(self):
        return self.d_res.ptr, 1 + len(self.ops)           # examples/fixed_point.rs:64, :79, :94, :110: make_public.push after each

    def results(self):
        return self.d_res.download((1 + len(self.ops), 4))[1:]
