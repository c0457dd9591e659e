"""N > 1 plumbing on the CPU (gloo, world_size 2): column sharding covers every column exactly once and
the commitment all_gather reassembles the per-rank [advice block | lookup block] shards in the unsharded
[all advice | all lookup] column order.  No GPU compute here."""
import os
import socket

import numpy as np
import pytest


def _worker(rank, world, port, n_adv, n_lk, q):
    import torch.distributed as dist
    from halo2_vectordb_amd.pipeline import column_shards, gather_commitments
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    (a_lo, a_hi), (l_lo, l_hi) = column_shards(n_adv, n_lk, world)[rank]
    mine = list(range(a_lo, a_hi)) + [n_adv + c for c in range(l_lo, l_hi)]     # global column numbers, buffer order
    local = (np.array(mine, dtype=np.uint64)[:, None] * np.uint64(8) + np.arange(8, dtype=np.uint64)[None, :]) * np.uint64(0x9E3779B97F4A7C15)
    out = gather_commitments(dist, local.reshape(len(mine), 8), column_shards(n_adv, n_lk, world), "cpu")
    q.put((rank, mine, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_adv,n_lk", [(7, 0), (1, 1), (8146, 25), (3, 5)])
def test_shard_and_gather_world2(n_adv, n_lk):
    n_cols = n_adv + n_lk
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_adv, n_lk, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(n_cols * 8, dtype=np.uint64).reshape(n_cols, 8) * np.uint64(0x9E3779B97F4A7C15)
    covered = []
    for rank, mine, out in sorted(res, key=lambda r: r[0]):
        assert np.array_equal(out, want)
        covered += mine
    assert sorted(covered) == list(range(n_cols))


def test_shard_ranges_partition():
    from halo2_vectordb_amd.pipeline import shard_range
    for n in (1, 5, 8146):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_balanced_shards_partition_and_balance():
    from halo2_vectordb_amd.pipeline import balanced_column_shards, balanced_ranges
    rng = np.random.default_rng(3)
    for n_adv, n_lk, w in ((1, 0, 2), (7, 3, 2), (100, 17, 3), (6801, 1345, 8)):
        va = rng.integers(0, 65536, n_adv) * (rng.random(n_adv) > 0.2)   # some nearly empty columns
        vl = rng.integers(60000, 65536, n_lk)
        shards = balanced_column_shards(va, vl, w)
        assert len(shards) == w
        for which, n in ((0, n_adv), (1, n_lk)):
            spans = [s[which] for s in shards]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(spans, spans[1:]))
    cost = np.r_[np.ones(500), 3 * np.ones(500)]
    spans = balanced_ranges(cost, 4)
    sums = [cost[a:b].sum() for a, b in spans]
    assert max(sums) - min(sums) <= 3.0 + 1e-9


def _partials_worker(rank, world, port, n_cols, q):
    import torch.distributed as dist
    from halo2_vectordb_amd.pipeline import allgather_partials, point_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = (np.arange(n_cols * 8, dtype=np.uint64).reshape(n_cols, 8) + np.uint64(1000 * rank)) * np.uint64(0x9E3779B97F4A7C15)
    parts = allgather_partials(dist, local, "cpu")
    q.put((rank, point_shard(1 << 16, rank, world), parts))
    dist.barrier()
    dist.destroy_process_group()


def test_point_sharded_partials_allgather_world2():
    """the exchange step of the point-sharded MSM (the alternative partition): every rank ends up with all ranks'
    partial commitments in rank order; the row slices partition the rows.  (The group sums are GPU work: tests/test_gpu_msm.py.)"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_partials_worker, args=(r, 2, port, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([(np.arange(24, dtype=np.uint64).reshape(3, 8) + np.uint64(1000 * r)) * np.uint64(0x9E3779B97F4A7C15) for r in range(2)])
    assert all(np.array_equal(parts, want) for _, _, parts in res)
    assert [sl for _, sl, _ in res] == [(0, 1 << 15), (1 << 15, 1 << 16)]


# ---------------------------------------------------------------------------------------------------------------------------
# the sharded prover rounds' bookkeeping (shardmap.ShardMap, pipeline.align_column_shards) and exchange steps (dist.Comm)
def test_aligned_shards_and_the_shard_map_tile_columns_and_sets():
    """For every rank count: the aligned blocks still tile the columns in rank order, every inner cut lies on a boundary of the
    permutation's sets, the ranks' sets tile [0, n_sets), the only columns a rank needs from another are the lookup columns that
    complete the set spanning the advice / lookup junction (and their holder has them as strays), and every requested boundary
    product belongs to another rank."""
    from halo2_vectordb_amd.pipeline import align_column_shards, balanced_column_shards, column_shards
    from halo2_vectordb_amd.shardmap import ShardMap
    rng = np.random.default_rng(5)
    for n_adv, n_lk in ((100, 20), (101, 20), (102, 7), (5, 0), (4, 2), (20, 1), (6801, 1345), (14164, 6805), (216, 49)):
        for world in (1, 2, 3, 4, 8):
            if world > n_adv:
                continue
            for balanced, chunk in ((False, 2), (True, 2), (False, 3), (True, 3)):       # chunk 2: cs.degree() - 2 of the reference's circuits
                if balanced:
                    shards = balanced_column_shards(rng.integers(0, 65536, n_adv), rng.integers(0, 65536, n_lk), world)
                else:
                    shards = column_shards(n_adv, n_lk, world)
                if world > 1:
                    before = shards
                    shards = align_column_shards(shards, n_adv, n_lk, chunk)
                    for (a0, l0), (a1, l1) in zip(before, shards):
                        assert abs(a0[1] - a1[1]) <= 2 and (abs(l0[1] - l1[1]) <= 2 or l1[1] in (0, n_lk) or l1[1] == (-n_adv) % chunk)
                m = ShardMap(shards, n_adv, n_lk, chunk)
                assert m.n_perm == n_adv + n_lk + 2 and m.n_sets == -(-m.n_perm // chunk)
                sets = sorted(x for r in range(world) for x in m.set_ranges(r))
                assert sets[0][0] == 0 and sets[-1][1] == m.n_sets and all(a[1] == b[0] for a, b in zip(sets, sets[1:]))
                held = sorted(p for r in range(world) for lo, hi in m.held_ranges(r) for p in range(lo, hi)) if n_adv < 1000 else None
                assert held is None or held == list(range(n_adv + n_lk))
                foreign = {r: m.foreign_cols(r) for r in range(world)}
                stray = {r: m.stray_cols(r) for r in range(world)}
                assert sorted(p for v in foreign.values() for p in v) == sorted(p for v in stray.values() for p in v)
                for r, cols in foreign.items():
                    assert len(cols) <= 2 and all(n_adv <= p < n_adv + 2 for p in cols)          # the junction's lookup columns only
                    assert all(m.col_owner(p) != r for p in cols)
                for r in range(world):
                    assert all(m.set_owner(i) != r for i in m.z_requests(r))
                    assert len(m.set_ranges(r)) <= 2
                if world == 1:
                    assert m.set_ranges(0) == [(0, m.n_sets)] and not foreign[0] and not stray[0] and not m.z_requests(0)
    with pytest.raises(ValueError):
        ShardMap(column_shards(100, 20, 2), 100, 20, 3)           # unaligned blocks: (0, 50) | (50, 100)


def _comm_worker(rank, world, port, q):
    import torch.distributed as dist
    from halo2_vectordb_amd.dist import Comm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = Comm(dist)
    assert (comm.rank, comm.world, comm.on_gpu) == (rank, world, False)
    # values only one rank holds, full-width words included (the sum is an integer all-reduce: exact with one contributor)
    whole = (np.arange(40, dtype=np.uint64).reshape(10, 4) + np.uint64(1)) * np.uint64(0xFFFFFFFFFFFFFFC5)
    mine = np.zeros_like(whole)
    mine[rank::world] = whole[rank::world]
    got = comm.sum_disjoint(mine)
    rows = comm.gather_rows(np.array([rank, 0xFFFFFFFFFFFFFFFF - rank], dtype=np.uint64))
    q.put((rank, bool(np.array_equal(got, whole)), rows.tolist(), comm.sum_disjoint(np.zeros((0, 4), dtype=np.uint64)).shape))
    comm.barrier()
    dist.destroy_process_group()


def test_comm_exchanges_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, rows, empty in res:
        assert ok and rows == [[0, 0xFFFFFFFFFFFFFFFF], [1, 0xFFFFFFFFFFFFFFFE]] and empty == (0, 4)
