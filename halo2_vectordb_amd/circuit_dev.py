"""The fixed-point circuits' constraint map assembled ON THE DEVICE: the block builders of circuit_sym.py (build_kmeans,
build_nearest) run unchanged with `DeviceBuilder` in place of the numpy `_Builder` — every `place` of a traced unit block at its
hundreds of thousands of stream offsets is one kernel (vdb_copymap_place_dev) instead of numpy fancy indexing over 10^9 cells, and
the four big arrays (copy_of, const_idx, flags, lookup_src) never exist on the host.  rounds.ProverRounds.keygen hands them to the
device-side cycle construction and MockProver as they are.

`DeviceCopyMap` has circuit_sym.CopyMap's surface; its array attributes download on first use (tests, small circuits)."""
import ctypes

import numpy as np

from . import api
from ._lib import check
from .circuit_sym import R, CopyMap


def _u64(v):
    return ctypes.c_uint64(int(v))


class DeviceCopyMap(CopyMap):
    def __init__(self, n_cells, n_lookup, d_copy_of, d_const_idx, d_flags, d_lookup_src, consts):
        self._n_cells, self.n_lookup = int(n_cells), int(n_lookup)
        self.d_copy_of, self.d_const_idx, self.d_flags, self.d_lookup_src = d_copy_of, d_const_idx, d_flags, d_lookup_src
        self.consts = consts
        self._host = {}

    @property
    def n_cells(self):
        return self._n_cells

    def _get(self, name):
        if name not in self._host:
            if self.d_copy_of is None:
                raise RuntimeError("the circuit's constraint map was released after keygen (ProverRounds.keep_circuit = True keeps it)")
            if name == "copy_of":
                self._host[name] = self.d_copy_of.download((self._n_cells,), dtype=np.int64)
            elif name == "const_idx":
                self._host[name] = self.d_const_idx.download((self._n_cells,), dtype=np.int64)
            elif name == "lookup_src":
                self._host[name] = self.d_lookup_src.download((self.n_lookup,), dtype=np.int64) if self.n_lookup else np.zeros(0, dtype=np.int64)
            else:
                f = self.d_flags.download((self._n_cells,), dtype=np.uint8)
                self._host["gate"], self._host["asserted"] = (f & 1).astype(bool), (f & 2).astype(bool)
        return self._host[name]

    copy_of = property(lambda self: self._get("copy_of"))
    const_idx = property(lambda self: self._get("const_idx"))
    lookup_src = property(lambda self: self._get("lookup_src"))
    gate = property(lambda self: self._get("gate"))
    asserted = property(lambda self: self._get("asserted"))

    def free(self):
        for name in ("d_copy_of", "d_const_idx", "d_flags", "d_lookup_src"):
            b = getattr(self, name)
            if b is not None:
                b.free()
                setattr(self, name, None)


class DeviceBuilder:
    """circuit_sym._Builder's interface (const_index, constant_cell, place, finish) over device arrays"""

    def __init__(self, n_cells, n_lookup):
        self.lib = api.init()
        self.n_cells, self.n_lookup = int(n_cells), int(n_lookup)
        self.d_copy_of = api.DeviceBuffer(max(self.n_cells, 1) * 8)
        self.d_const_idx = api.DeviceBuffer(max(self.n_cells, 1) * 8)
        self.d_flags = api.DeviceBuffer(max(self.n_cells, 1))
        self.d_lookup_src = api.DeviceBuffer(max(self.n_lookup, 1) * 8)
        check(self.lib.vdb_copymap_init_dev(_u64(self.n_cells), _u64(self.n_lookup), self.d_copy_of.ptr, self.d_const_idx.ptr, self.d_flags.ptr, self.d_lookup_src.ptr))
        self.consts, self._cmap = [], {}
        self._blocks = {}          # id(block) -> its template on the device
        self._small = api.DeviceBuffer(1 << 20)

    def const_index(self, v):
        if v not in self._cmap:
            self._cmap[v] = len(self.consts)
            self.consts.append(v)
        return self._cmap[v]

    def constant_cell(self, pos, value):
        self.d_const_idx.upload(np.array([self.const_index(value % R)], dtype=np.int64), offset=int(pos) * 8)

    def _template(self, blk):
        key = id(blk)
        if key not in self._blocks:
            remap = np.asarray([self.const_index(v) for v in blk.consts], dtype=np.int64)
            cid = np.where(blk.cidx >= 0, remap[np.maximum(blk.cidx, 0)] if len(remap) else -1, -1).astype(np.int64)
            flags = blk.gate.astype(np.uint8)
            if blk.asserted.size:
                flags[blk.asserted] |= 2
            arrays = [np.ascontiguousarray(blk.src, dtype=np.int64), cid, flags, np.ascontiguousarray(blk.lk, dtype=np.int64)]
            bufs = []
            for a in arrays:
                b = api.DeviceBuffer(max(a.nbytes, 32))
                if a.nbytes:
                    b.upload(a)
                bufs.append(b)
            self._blocks[key] = (blk, bufs)          # (the block is kept alive: its id is the key)
        return self._blocks[key][1]

    def place(self, blk, bases, lk_bases, ext_cells):
        bases = np.ascontiguousarray(bases, dtype=np.int64).reshape(-1)
        m = bases.size
        ext_cells = np.ascontiguousarray(ext_cells, dtype=np.int64).reshape(m, -1)
        n_ext = ext_cells.shape[1]
        lkb = np.ascontiguousarray(lk_bases, dtype=np.int64).reshape(-1) if blk.n_lk else np.zeros(0, dtype=np.int64)
        src, cid, flags, lk = self._template(blk)
        need = bases.nbytes + lkb.nbytes + ext_cells.nbytes + 64
        if need > self._small.nbytes:
            self._small.free()
            self._small = api.DeviceBuffer(need * 2)
        o1 = (bases.nbytes + 7) // 8 * 8
        o2 = o1 + (lkb.nbytes + 7) // 8 * 8
        if m:
            self._small.upload(bases)
            if lkb.nbytes:
                self._small.upload(lkb, offset=o1)
            if ext_cells.nbytes:
                self._small.upload(ext_cells, offset=o2)
            check(self.lib.vdb_copymap_place_dev(src.ptr, cid.ptr, flags.ptr, _u64(blk.n), lk.ptr, _u64(blk.n_lk), self._small.ptr, self._small.at(o1), self._small.at(o2),
                                                 _u64(m), _u64(n_ext), _u64(self.n_cells), _u64(self.n_lookup), self.d_copy_of.ptr, self.d_const_idx.ptr,
                                                 self.d_flags.ptr, self.d_lookup_src.ptr))
        return bases[:, None] + blk.outs[None, :]

    def finish(self):
        bad, nosrc = ctypes.c_uint64(), ctypes.c_uint64()
        check(self.lib.vdb_copymap_finish_dev(self.d_copy_of.ptr, self.d_const_idx.ptr, _u64(self.n_cells), self.d_lookup_src.ptr, _u64(self.n_lookup), None,
                                              ctypes.byref(bad), ctypes.byref(nosrc)))
        assert nosrc.value == 0, "lookup cells without a source"
        for _blk, bufs in self._blocks.values():
            for b in bufs:
                b.free()
        self._small.free()
        return DeviceCopyMap(self.n_cells, self.n_lookup, self.d_copy_of, self.d_const_idx, self.d_flags, self.d_lookup_src, self.consts)


# ---------------------------------------------------------------------------------------------------------------------------
# merkle_commitment (copymap.py's map) placed by the device: one block per kind of permutation
class _PermBlock:
    """one PoseidonChip::permutation as a unit block for DeviceBuilder.place: copymap.permutation_template's cell codes turned into
    the builder's (external inputs 0..2: the sponge state, 3..4: the absorbed words), for one combination of
    (words absorbed, state starts from the chip's initial state, the words copy assigned cells)"""

    def __init__(self, src, fin, flags, values, fresh, tied_msgs):
        from .circuit_sym import EXT0, SELF
        from .copymap import T
        src = np.asarray(src, dtype=np.int64)
        self.n, self.n_lk = src.size, 0
        out = np.where(src >= 0, src, SELF).astype(np.int64)
        cst = (np.asarray(flags) & 2) != 0
        consts, cmap = [], {}
        cidx = np.full(src.size, -1, dtype=np.int64)

        def cid(v):
            if v not in cmap:
                cmap[v] = len(consts)
                consts.append(v)
            return cmap[v]
        for i in np.flatnonzero(cst):
            cidx[i] = cid(int(values[i]))
        for i in range(T):
            cols = np.flatnonzero(src == -10 - i)
            if fresh:                                  # capacity 2^64, then zeros: constants of the circuit though the kernels do not flag them
                for c in cols:
                    cidx[c] = cid((1 << 64) if i == 0 else 0)
            else:
                out[cols] = EXT0 - i
        for i in range(2):
            cols = np.flatnonzero(src == -20 - i)
            if tied_msgs:
                out[cols] = EXT0 - (T + i)
        self.src, self.cidx, self.consts = out, cidx, consts
        self.gate = (np.asarray(flags) & 1).astype(bool)
        self.asserted = np.zeros(0, dtype=np.int64)
        self.lk = np.zeros(0, dtype=np.int64)
        self.outs = np.asarray(fin, dtype=np.int64)


def place_merkle(B, n, dim, base, vec_base, fetch_flags, fetch_values):
    """merkle_commitment over n vectors of `dim` words (src/gadget/vectordb.rs:165-223) whose trace starts at stream cell `base`,
    placed into DeviceBuilder `B` in the cell order of witness.hip: the leaves' sponges, the load_zero cell of the padding, the
    tree.  `vec_base`: stream cell of word 0 of vector 0 (the assigned vectors the leaves absorb; None: free words).
    fetch_flags(lo, hi) / fetch_values(lo, hi): the kernel's flag bytes / the canonical values of stream cells [lo, hi) of a
    keygen-style run (one instance of each kind of permutation is read).  Returns the stream cell of the root."""
    from .copymap import T, perm_cells, permutation_template
    nperm = (dim + 1) // 2 + (1 if dim % 2 == 0 else 0)
    n_ins = [max(0, min(2, dim - 2 * p)) for p in range(nperm)]
    sizes = [perm_cells(k) for k in n_ins]
    leaf_cells = sum(sizes)
    lp = 1
    while lp < n:
        lp <<= 1
    zero_cell = 1 if lp > n else 0
    node_cells = perm_cells(2) + perm_cells(0)
    blocks = {}

    def block(n_in, at, fresh, tied):
        key = (n_in, fresh, tied)
        if key not in blocks:
            size = perm_cells(n_in)
            flags = np.asarray(fetch_flags(at, at + size), dtype=np.uint8)
            src, fin = permutation_template(flags, n_in)
            blocks[key] = _PermBlock(src, fin, flags, fetch_values(at, at + size), fresh, tied)
        return blocks[key]

    def place(bases, n_in, state, msgs):
        m = bases.size
        ext = np.zeros((m, T + 2), dtype=np.int64)
        fresh = state[0] is None
        if not fresh:
            for i in range(T):
                ext[:, i] = state[i]
        tied = n_in > 0 and msgs[0] is not None
        if tied:
            for i in range(n_in):
                ext[:, T + i] = msgs[i]
        outs = B.place(block(n_in, int(bases[0]), fresh, tied), bases, np.zeros(m, dtype=np.int64), ext)
        return [outs[:, i] for i in range(T)]

    leaf_base = base + np.arange(n, dtype=np.int64) * leaf_cells
    state, off = [None] * T, 0
    for p in range(nperm):
        msg = [vec_base + np.arange(n, dtype=np.int64) * dim + 2 * p + i for i in range(n_ins[p])] if vec_base is not None else [None] * n_ins[p]
        state = place(leaf_base + off, n_ins[p], state, msg)
        off += sizes[p]
    zero_at = base + n * leaf_cells
    if zero_cell:
        B.constant_cell(zero_at, 0)                          # ctx.load_zero() for the padding leaves
    digest = np.full(lp, zero_at, dtype=np.int64)
    digest[:n] = state[1]
    pos, width = base + n * leaf_cells + zero_cell, lp
    while width > 1:
        half = width // 2
        bases = pos + np.arange(half, dtype=np.int64) * node_cells
        st1 = place(bases, 2, [None] * T, [digest[0:width:2], digest[1:width:2]])
        st2 = place(bases + perm_cells(2), 0, st1, [])
        digest = st2[1]
        pos += half * node_cells
        width = half
    return int(digest[0]), pos
