"""Who holds what in the sharded prover rounds (rounds.py), derived from the hot path's column blocks — plain integer
bookkeeping, the same on every rank.

The permutation argument runs over the columns [advice | lookup | constants | instance] in sets of `chunk` consecutive
columns, one running-product polynomial per set.  A rank holds a block of the advice columns and a block of the lookup
columns (pipeline.column_shards / balanced_column_shards, cut on set boundaries by pipeline.align_column_shards); a set
belongs to the rank that holds its FIRST column (the constants' and the instance column count as the last rank's).  So a
rank's sets are at most two contiguous ranges — those that start in its advice block and those that start in its lookup
block — and the only columns a rank needs without holding them are the lookup columns that complete the one set spanning
the advice / lookup junction ("foreign" columns, at most chunk - 1; their holder in turn has them outside its own sets:
"stray" columns).  Everything else a rank needs from another is one boundary polynomial per range (the previous range's
last running product, for the term that chains the products) and the first product for the rank that closes the chain.
"""


def _ceil_div(a, b):
    return -(-a // b)


class ShardMap:
    def __init__(self, shards, n_adv, n_lk, chunk):
        self.shards, self.n_adv, self.n_lk, self.chunk = [tuple(map(tuple, s)) for s in shards], n_adv, n_lk, chunk
        self.world = len(shards)
        self.n_cols = n_adv + n_lk
        self.n_perm = self.n_cols + 2
        self.n_sets = _ceil_div(self.n_perm, chunk)
        a_end = l_end = 0
        for (a_lo, a_hi), (l_lo, l_hi) in self.shards:
            if a_lo != a_end or l_lo != l_end or a_hi < a_lo or l_hi < l_lo:
                raise ValueError("column blocks must tile the columns in rank order")
            a_end, l_end = a_hi, l_hi
            if self.world > 1 and (a_lo % chunk or (l_lo and (n_adv + l_lo) % chunk)):
                raise ValueError("column blocks must start on the permutation's set boundaries (pipeline.align_column_shards)")
        if a_end != n_adv or l_end != n_lk:
            raise ValueError("column blocks must tile the columns in rank order")
        first_lk_set = _ceil_div(n_adv, chunk)
        self._ranges = []
        for r, ((a_lo, a_hi), (l_lo, l_hi)) in enumerate(self.shards):
            adv = (a_lo // chunk, _ceil_div(a_hi, chunk)) if a_hi > a_lo else None
            lo = max(first_lk_set, _ceil_div(n_adv + l_lo, chunk))
            hi = self.n_sets if r == self.world - 1 else max(lo, _ceil_div(n_adv + l_hi, chunk))
            lk = (lo, hi) if hi > lo else None
            if adv and lk and adv[1] == lk[0]:
                rs = [(adv[0], lk[1])]
            else:
                rs = [x for x in (adv, lk) if x]
            self._ranges.append(rs)
        covered = sorted(x for rs in self._ranges for x in rs)
        pos = 0
        for lo, hi in covered:
            if lo != pos:
                raise ValueError("the ranks' sets do not tile the permutation's sets")
            pos = hi
        if pos != self.n_sets:
            raise ValueError("the ranks' sets do not tile the permutation's sets")

    # ------------------------------------------------------------------ columns
    def col_owner(self, p):
        """rank that holds permutation column p (an advice or lookup column; the constants' and the instance column are
        everyone's: None)"""
        if p >= self.n_cols:
            return None
        for r, ((a_lo, a_hi), (l_lo, l_hi)) in enumerate(self.shards):
            if a_lo <= p < a_hi or l_lo <= p - self.n_adv < l_hi:
                return r
        raise ValueError("column without a holder")

    def held_ranges(self, rank):
        """the permutation columns rank holds, as ranges in permutation numbering: [advice block, lookup block]"""
        (a_lo, a_hi), (l_lo, l_hi) = self.shards[rank]
        return [(a_lo, a_hi), (self.n_adv + l_lo, self.n_adv + l_hi)]

    # ------------------------------------------------------------------ sets
    def set_ranges(self, rank):
        """[(s_lo, s_hi)]: the sets rank computes the running products of, in increasing order (one or two ranges)"""
        return list(self._ranges[rank])

    def all_ranges(self):
        """[(s_lo, s_hi, rank)] of every rank, in the order of the sets"""
        return sorted((lo, hi, r) for r, rs in enumerate(self._ranges) for lo, hi in rs)

    def set_owner(self, i):
        for lo, hi, r in self.all_ranges():
            if lo <= i < hi:
                return r
        raise ValueError("set without an owner")

    def range_cols(self, rng):
        """permutation columns of a range of sets"""
        return rng[0] * self.chunk, min(rng[1] * self.chunk, self.n_perm)

    def set_col_ranges(self, rank):
        """the permutation columns of rank's sets (the sigma columns it keeps), as ranges"""
        return [self.range_cols(rng) for rng in self._ranges[rank]]

    def foreign_cols(self, rank):
        """advice / lookup columns inside rank's sets that another rank holds"""
        return [p for lo, hi in self.set_col_ranges(rank) for p in range(lo, min(hi, self.n_cols)) if self.col_owner(p) != rank]

    def stray_cols(self, rank):
        """columns rank holds that lie in no set of its own"""
        mine = self.set_col_ranges(rank)
        return [p for lo, hi in self.held_ranges(rank) for p in range(lo, hi) if not any(a <= p < b for a, b in mine)]

    def z_requests(self, rank):
        """product polynomials of other ranks that rank needs in coefficient form: the set before each of its ranges (the
        chaining term l0 (z_i - z_{i-1}(w^-b X))), and set 0 when it closes the chain (l_last (z_last^2 - z_last) is folded in
        together with l0 (1 - z_0))"""
        need = set()
        for lo, _hi in self._ranges[rank]:
            if lo > 0:
                need.add(lo - 1)
        if self._ranges[rank] and self.head_owner() == rank and self.set_owner(0) != rank:
            need.add(0)
        return sorted(i for i in need if self.set_owner(i) != rank)

    def head_owner(self):
        """the rank that folds in the two terms that read only the first and the last product: the owner of the last set"""
        return self.set_owner(self.n_sets - 1)

    def all_foreign_cols(self):
        return sorted({p for r in range(self.world) for p in self.foreign_cols(r)})

    def all_z_requests(self):
        return sorted({i for r in range(self.world) for i in self.z_requests(r)})
