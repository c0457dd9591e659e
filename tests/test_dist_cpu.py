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
