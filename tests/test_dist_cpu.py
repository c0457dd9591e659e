"""N > 1 plumbing on the CPU (gloo, world_size 2): column sharding covers every column exactly once and
the commitment all_gather reassembles the per-rank shards in column order.  No GPU compute here."""
import os
import socket

import numpy as np
import pytest


def _worker(rank, world, port, n_cols, q):
    import torch.distributed as dist
    from halo2_vectordb_amd.pipeline import gather_commitments, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_cols, rank, world)
    local = np.arange(lo * 8, hi * 8, dtype=np.uint64).reshape(hi - lo, 8) * np.uint64(0x9E3779B97F4A7C15)
    out = gather_commitments(dist, local, n_cols, rank, world, "cpu")
    q.put((rank, lo, hi, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_cols", [7, 8146])
def test_shard_and_gather_world2(n_cols):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_cols, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(n_cols * 8, dtype=np.uint64).reshape(n_cols, 8) * np.uint64(0x9E3779B97F4A7C15)
    covered = []
    for rank, lo, hi, out in sorted(res, key=lambda r: r[0]):
        assert np.array_equal(out, want)
        covered += list(range(lo, hi))
    assert covered == list(range(n_cols))


def test_shard_ranges_partition():
    from halo2_vectordb_amd.pipeline import shard_range
    for n in (1, 5, 8146):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
