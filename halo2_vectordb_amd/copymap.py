"""The Merkle circuit's own copy constraints (what keygen feeds the permutation argument beyond the cells
the column layout duplicates): for every cell of the Poseidon trace that `merkle_commitment` emits
(/root/reference/src/gadget/vectordb.rs:165-223 through PoseidonChip<F, 3, 2>), the earlier cell it is a copy of.

The trace is data independent, so the map is built symbolically on the host from the cell templates of the halo2-base
primitives the permutation is made of — GateChip::sum, inner_product with constants, mul, mul_add — in exactly the order
halo2_vectordb_amd/csrc/witness.hip (trace_permutation) emits them, and instantiated per permutation with numpy.  The
builder consumes the gate / constant flag bytes of one real permutation (a keygen-style run) both to resolve the one
structural choice that depends on the spec's constants (inner_product starts with the first operand itself when its
constant is one) and as a check: every gate-start bit and every constant bit of the template must equal the kernel's.
tests/test_gpu_rounds.py additionally checks the map against witness values: every cell equals the cell it copies.

The leaves' message words are copies of the vector cells assigned at the head of the stream (ctx.assign_witnesses before
the gadget runs, as the reference's chip_merkle does).  Constant cells are not part of the map: they are pinned by
the constants gate of the rounds (rounds.py) instead of being tied to a fixed column.  [UPSTREAM-RECALL] for the primitives' cell templates, as for the kernels themselves."""
import numpy as np

T, RATE, R_F, R_P = 3, 2, 8, 57
HALF = R_F // 2
SELF, CONST = -1, -2          # template codes; >= 0: offset inside the block; state input i: -10 - i; message input i: -20 - i


class _Tracer:
    """symbolic WCtx: records per cell (source code, gate bit, constant bit)"""

    def __init__(self, flags):
        self.src, self.gate, self.cst, self.flags = [], [], [], flags

    def push(self, src, gate, cst=False):
        pos = len(self.src)
        self.src.append(CONST if cst else (SELF if src is None else src))
        self.gate.append(bool(gate))
        self.cst.append(bool(cst))
        return pos

    def next_is_const(self):
        return bool(self.flags[len(self.src)] & 2)

    # GateChip::sum: v0, then (v_i, 1, s_i); `cmask` bit i: v_i is a constant
    def sum(self, v, cmask):
        self.push(v[0], len(v) > 1, bool(cmask & 1))
        out = None
        for i in range(1, len(v)):
            self.push(v[i], False, bool((cmask >> i) & 1))
            self.push(None, False, True)
            out = self.push(None, i + 1 < len(v))
        return out

    # inner_product(a, constants): starts with a_0 itself when the first constant is one, with a constant zero otherwise
    def ip_const(self, a):
        n = len(a)
        if self.next_is_const():
            self.push(None, True, True)
            i0, ng = 0, n
        else:
            self.push(a[0], n - 1 > 0)
            i0, ng = 1, n - 1
        out, gi = None, 1
        for i in range(i0, n):
            self.push(a[i], False)
            self.push(None, False, True)
            out = self.push(None, gi < ng)
            gi += 1
        return out

    def mul(self, a, b):                       # [0, a, b, out]
        self.push(None, True, True)
        self.push(a, False)
        self.push(b, False)
        return self.push(None, False)

    def sbox(self, x):                         # x^5 + c: mul(x, x), mul(x2, x2), mul_add(x, x4, c) = [c, x, x4, out]
        x2 = self.mul(x, x)
        x4 = self.mul(x2, x2)
        self.push(None, True, True)
        self.push(x, False)
        self.push(x4, False)
        return self.push(None, False)

    def dense(self, st):
        return [self.ip_const(st) for _ in range(T)]


def permutation_template(flags, n_in):
    """(src codes, final state offsets) of one PoseidonChip::permutation absorbing n_in message words; `flags`: the kernel's
    flag bytes of one such permutation (bit 0 gate start, bit 1 constant)."""
    t = _Tracer(flags)
    st = [-10 - i for i in range(T)]
    st[0] = t.sum([st[0], None], 0b10)
    for i in range(n_in):
        st[1 + i] = t.sum([st[1 + i], -20 - i, None], 0b100)
    for i in range(n_in + 1, T):
        st[i] = t.sum([st[i], None], 0b10)
    for _ in range(1, HALF):
        st = t.dense([t.sbox(x) for x in st])
    st = t.dense([t.sbox(x) for x in st])                       # last of the first full rounds, then the pre-sparse matrix
    for _ in range(R_P):
        s0 = t.sbox(st[0])
        cur = [s0, st[1], st[2]]
        nxt = [t.ip_const(cur)]
        for i in range(1, T):                                   # mul_add(s0, e, s_i) = [s_i, s0, e, out]
            t.push(cur[i], True)
            t.push(s0, False)
            t.push(None, False, True)
            nxt.append(t.push(None, False))
        st = nxt
    for _ in range(HALF - 1):
        st = t.dense([t.sbox(x) for x in st])
    st = t.dense([t.sbox(x) for x in st])
    n = len(t.src)
    if n != len(flags):
        raise ValueError(f"permutation template has {n} cells, the kernel emitted {len(flags)}")
    gate = np.array(t.gate, dtype=bool)
    cst = np.array(t.cst, dtype=bool)
    f = np.asarray(flags)
    if not (np.array_equal(gate, (f & 1).astype(bool)) and np.array_equal(cst, (f & 2).astype(bool))):
        raise ValueError("permutation template disagrees with the kernel's gate / constant flags")
    return np.array(t.src, dtype=np.int64), st


def perm_cells(n_in):
    return {2: 18, 1: 15, 0: 12}[n_in] + 2238


def merkle_copy_map(n, dim, flags, vectors_assigned=True, placements=None):
    """copy_of[i] = stream offset of the cell that cell i copies (i itself for new cells and constants) for merkle_commitment
    over n vectors of `dim` words; `flags`: the flag bytes of a keygen-style run of the same circuit (vdb_wit_merkle_dev with a
    selector buffer).  With `vectors_assigned` the stream starts with the n * dim assigned vector words
    (ctx.assign_witnesses, as the reference's chip_merkle does) and every message word a leaf absorbs is a copy of its
    cell; without, the message words are free cells.  Also returns the stream offset of the root cell and the offsets of the
    cells that hold the sponge's initial state (every leaf and every tree node starts from [2^64, 0, 0]; the kernels emit them
    as ordinary cells, the circuit must pin them like constants)."""
    flags = np.asarray(flags, dtype=np.uint8)
    n_in = n * dim if vectors_assigned else 0
    if n_in:
        if flags[:n_in].any():
            raise ValueError("the assigned vector words carry no gate or constant flag")
        inner = [] if placements is not None else None
        gadget, root, init = merkle_copy_map(n, dim, flags[n_in:], vectors_assigned=None, placements=inner)
        if placements is not None:
            placements.extend((k, bases + n_in, states) for k, bases, states in inner)
        return np.concatenate([np.arange(n_in, dtype=np.int64), gadget + n_in]), root + n_in, init + n_in
    nperm = (dim + 1) // 2 + (1 if dim % 2 == 0 else 0)
    n_ins = [max(0, min(2, dim - 2 * p)) for p in range(nperm)]
    sizes = [perm_cells(k) for k in n_ins]
    leaf_cells = sum(sizes)
    lp = 1
    while lp < n:
        lp <<= 1
    zero_cell = 1 if lp > n else 0
    node_cells = perm_cells(2) + perm_cells(0)
    total = n * leaf_cells + zero_cell + (lp - 1) * node_cells
    if total != flags.size:
        raise ValueError("flags do not belong to this circuit")
    copy_of = np.arange(total, dtype=np.int64)
    templates = {}
    init_cells = []          # cells that hold the sponge's initial state [2^64, 0, 0]: constants of the circuit, though not flagged

    def template(n_in, at):
        if n_in not in templates:
            templates[n_in] = permutation_template(flags[at: at + perm_cells(n_in)], n_in)
        return templates[n_in]

    def place(bases, n_in, state_src, msg_src):
        """instantiate the template at every offset in `bases`; state_src / msg_src: per input, array of source offsets (or
        None: a free / constant-initialised cell)"""
        src, fin = template(n_in, int(bases[0]))
        if placements is not None:      # (message words absorbed, instance offsets, which of the T state inputs start from the chip's initial state)
            placements.append((n_in, bases.copy(), [st is None for st in state_src]))
        idx = bases[:, None] + np.arange(src.size)[None, :]
        val = np.where(src[None, :] >= 0, bases[:, None] + np.maximum(src, 0)[None, :], idx)
        for i in range(T):
            cols = np.flatnonzero(src == -10 - i)
            if state_src[i] is not None and cols.size:
                val[:, cols] = state_src[i][:, None]
            elif cols.size:
                init_cells.append(idx[:, cols].reshape(-1))
        for i in range(n_in):
            cols = np.flatnonzero(src == -20 - i)
            if msg_src[i] is not None and cols.size:
                val[:, cols] = msg_src[i][:, None]
        copy_of[idx.reshape(-1)] = val.reshape(-1)
        return [bases + f for f in fin]

    # leaves: the sponge state runs through the leaf's permutations; the first starts from the chip's initial state
    leaf_base = np.arange(n, dtype=np.int64) * leaf_cells
    state, off = [None] * T, 0
    for p in range(nperm):
        # (vectors_assigned is None in the inner call of the assigned-vectors case: the words sit n * dim cells before the gadget)
        msg = [np.arange(n, dtype=np.int64) * dim + 2 * p + i - n * dim for i in range(n_ins[p])] if vectors_assigned is None else [None] * n_ins[p]
        state = place(leaf_base + off, n_ins[p], state, msg)
        off += sizes[p]
    digest = np.full(lp, n * leaf_cells, dtype=np.int64)        # padding leaves: the load_zero cell after the leaves
    digest[:n] = state[1]
    # tree levels: absorb [left, right], then the padding-only permutation
    pos, width = n * leaf_cells + zero_cell, lp
    while width > 1:
        half = width // 2
        bases = pos + np.arange(half, dtype=np.int64) * node_cells
        st1 = place(bases, 2, [None] * T, [digest[0:width:2], digest[1:width:2]])
        st2 = place(bases + perm_cells(2), 0, st1, [])
        digest = st2[1]
        pos += half * node_cells
        width = half
    return copy_of, int(digest[0]), np.sort(np.concatenate(init_cells))


def lookup_sources(flags, n_lookup):
    """For every cell of the lookup stream (cells_to_lookup: copies of advice cells, laid out in the lookup columns) the
    stream offset of the advice cell it copies, from a keygen-style run's flag bytes: the kernels mark those cells with bit 2
    in the order they queue them (gadgets.hpp r_range_check).  Raises when the counts differ — a circuit that looks a cell
    up that was assigned somewhere else (a range check of at most lookup_bits bits) is not covered."""
    src = np.flatnonzero(np.asarray(flags, dtype=np.uint8) & 4).astype(np.int64)
    if src.size != n_lookup:
        raise ValueError(f"{src.size} advice cells are marked as lookup sources for {n_lookup} lookup cells")
    return src


def mapping_from_copy_of(copy_of, break_points, n_cols, rows, lookup_src=None, lookup_rows=None, const_idx=None, n_consts=0, instance_cells=None):
    """Permutation over a grid of columns x rows (words col << 32 | row, the identity where nothing is tied) from a copy map
    over the stream cells that fill the first len(break_points) + 1 columns: every set of cells that copy one another
    (directly or through other copies) becomes one cycle, and the overlap cell that ends column c is the cell that starts
    column c + 1.  `lookup_src` (with `lookup_rows` cells per lookup column): lookup cell j, at row j % lookup_rows of column
    n_adv + j // lookup_rows, joins the cycle of the advice cell lookup_src[j].
    `const_idx` (with `n_consts`): cell i with const_idx[i] = r >= 0 is tied to row r of ONE MORE column, number n_cols, the
    fixed column that holds the circuit's constants — how halo2-base pins `QuantumCell::Constant` cells and
    `assert_is_const` — and the grid gets n_cols + 1 columns.
    `instance_cells` (a list, possibly empty; requires `const_idx`): the grid gets column n_cols + 1 as well, the instance
    column, whose row i joins the cycle of stream cell instance_cells[i] (RangeWithInstanceCircuitBuilder's constrain_instance)."""
    copy_of = np.asarray(copy_of, dtype=np.int64)
    n_cells = copy_of.size
    bp = np.asarray(break_points, dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(bp)])
    with_consts = const_idx is not None
    root = copy_of.copy()
    if with_consts:
        const_idx = np.asarray(const_idx, dtype=np.int64)
        tied = np.flatnonzero(const_idx >= 0)
        if (copy_of[tied] != tied).any():
            raise ValueError("a cell tied to a constant must be the root of its copies")
        root = np.concatenate([root, n_cells + np.arange(n_consts, dtype=np.int64)])      # the fixed column's cells follow the stream
        root[tied] = n_cells + const_idx[tied]
    while True:                                                 # pointer jumping: sources are earlier cells or fixed cells
        nxt = root[root]
        if np.array_equal(nxt, root):
            break
        root = nxt
    s = np.arange(n_cells, dtype=np.int64)
    col = np.searchsorted(starts, s, side="right") - 1
    row = s - starts[col]
    # the overlap cell: stream offset starts[c + 1] also sits in column c at row bp[c]
    dup_s = starts[1:][starts[1:] < n_cells]
    dup_col = np.arange(dup_s.size, dtype=np.int64)
    pos_col = [col, dup_col]
    pos_row = [row, bp[: dup_s.size]]
    pos_root = [root[:n_cells], root[dup_s]]
    if lookup_src is not None and len(lookup_src):
        j = np.arange(len(lookup_src), dtype=np.int64)
        pos_col.append(bp.size + 1 + j // lookup_rows)
        pos_row.append(j % lookup_rows)
        pos_root.append(root[np.asarray(lookup_src, dtype=np.int64)])
    total_cols = n_cols + (1 if with_consts else 0)
    if with_consts:
        if n_consts > rows:
            raise ValueError("more distinct constants than rows in the fixed column")
        pos_col.append(np.full(n_consts, n_cols, dtype=np.int64))
        pos_row.append(np.arange(n_consts, dtype=np.int64))
        pos_root.append(root[n_cells:])
    if instance_cells is not None:
        if not with_consts:
            raise ValueError("the instance column follows the constants' column")
        inst = np.asarray(instance_cells, dtype=np.int64).reshape(-1)
        if inst.size > rows:
            raise ValueError("more public cells than rows in the instance column")
        total_cols += 1
        pos_col.append(np.full(inst.size, n_cols + 1, dtype=np.int64))
        pos_row.append(np.arange(inst.size, dtype=np.int64))
        pos_root.append(root[inst])
    pos_col, pos_row, pos_root = np.concatenate(pos_col), np.concatenate(pos_row), np.concatenate(pos_root)
    if int(root.size) * total_cols * rows < (1 << 62):           # one combined key sorts several times faster than three
        order = np.argsort(pos_root * (total_cols * rows) + pos_col * rows + pos_row, kind="stable")
    else:
        order = np.lexsort((pos_row, pos_col, pos_root))
    pc, prw, pr = pos_col[order], pos_row[order], pos_root[order]
    first = np.concatenate([[True], pr[1:] != pr[:-1]])
    group_start = np.maximum.accumulate(np.where(first, np.arange(pr.size), 0))
    last = np.concatenate([pr[1:] != pr[:-1], [True]])
    nxt_idx = np.where(last, group_start, np.arange(pr.size) + 1)
    mapping = (np.arange(total_cols, dtype=np.uint64)[:, None] << np.uint64(32)) | np.arange(rows, dtype=np.uint64)[None, :]
    mapping[pc, prw] = (pc[nxt_idx].astype(np.uint64) << np.uint64(32)) | prw[nxt_idx].astype(np.uint64)
    return mapping


def merkle_circuit_map(n, dim, flags, fetch, vectors_assigned=True):
    """The Merkle circuit's whole constraint map as a circuit_sym.CopyMap, plus the root cell: merkle_copy_map's copies, the
    lookup-free gate flags, and every constant cell with the fixed-column value it is tied to — the cells the kernels flag
    (bit 1: Poseidon round constants, matrix entries, the ones and zeros of the gate templates), the zero cell of the padding,
    and the sponge's initial state [2^64, 0, 0] at the start of every leaf and tree node, which the kernels emit as ordinary
    cells.  Constants are data independent and repeat with the permutation template, so their values are read from ONE instance
    of each template through `fetch(lo, hi)` -> canonical integers of stream cells [lo, hi) of a keygen-style run."""
    from .circuit_sym import CopyMap
    flags = np.asarray(flags, dtype=np.uint8)
    placements = []
    copy_of, root, _init = merkle_copy_map(n, dim, flags, vectors_assigned, placements)
    const_idx = np.full(flags.size, -1, dtype=np.int64)
    consts, cmap = [], {}

    def cid(v):
        if v not in cmap:
            cmap[v] = len(consts)
            consts.append(v)
        return cmap[v]

    first = {}
    for n_in, bases, fresh_state in placements:
        size = perm_cells(n_in)
        if n_in not in first:
            b0 = int(bases[0])
            src, _ = permutation_template(flags[b0: b0 + size], n_in)
            vals = fetch(b0, b0 + size)
            cst = np.flatnonzero(flags[b0: b0 + size] & 2)
            first[n_in] = (src, cst, np.asarray([cid(int(vals[i])) for i in cst], dtype=np.int64))
        src, cst, ids = first[n_in]
        const_idx[(bases[:, None] + cst[None, :]).reshape(-1)] = np.broadcast_to(ids[None, :], (bases.size, cst.size)).reshape(-1)
        for i, fresh in enumerate(fresh_state):            # state word i starts from the chip's initial state: capacity 2^64, then zeros
            cols = np.flatnonzero(src == -10 - i)
            if fresh and cols.size:
                const_idx[(bases[:, None] + cols[None, :]).reshape(-1)] = cid((1 << 64) if i == 0 else 0)
    # flagged cells outside the permutations (the padding's load_zero cell)
    rest = np.flatnonzero(((flags & 2) != 0) & (const_idx < 0))
    for i in rest:
        const_idx[i] = cid(int(fetch(int(i), int(i) + 1)[0]))
    tied = const_idx >= 0
    if (copy_of[tied] != np.flatnonzero(tied)).any():
        raise ValueError("a constant cell of the Merkle circuit copies another cell")
    return CopyMap(copy_of, const_idx, consts, np.zeros(flags.size, dtype=bool), (flags & 1).astype(bool), np.zeros(0, dtype=np.int64)), root
