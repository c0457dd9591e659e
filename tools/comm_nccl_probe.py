#!/usr/bin/env python3
"""dist.Comm over the RCCL ("nccl") backend with the ranks this box has GPUs for (one on a one-GPU box): the exchange steps of the
sharded proof with their tensors on the device — the integer all-reduce of disjoint arrays, the gather of rows, and the field sum of
device buffers, whose partial sums travel GPU to GPU (all_gather_into_tensor on torch tensors filled by vdb_memcpy_d2d from the
library's own allocations).  Launched by tests/test_gpu_sharded.py; prints one JSON line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.dist import Comm
    api.init(dev)
    comm = Comm(dist)
    assert comm.on_gpu and (comm.rank, comm.world) == (rank, world)
    whole = (np.arange(64, dtype=np.uint64).reshape(16, 4) + np.uint64(1)) * np.uint64(0xFFFFFFFFFFFFFFC5)
    mine = np.zeros_like(whole)
    mine[rank::world] = whole[rank::world]
    ok_disjoint = bool(np.array_equal(comm.sum_disjoint(mine), whole))
    rows = comm.gather_rows(np.array([rank, 7], dtype=np.uint64))
    ok_rows = rows.shape == (world, 2) and [int(r[0]) for r in rows] == list(range(world))
    # field sum of device buffers: rank r holds the canonical values (i + 1) * (r + 1); the sum is (i + 1) * world (world + 1) / 2
    n = 1 << 14
    vals = api.fr_from_canonical(np.stack([(np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(rank + 1), np.zeros(n, dtype=np.uint64),
                                           np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)], axis=1))
    buf = api.DeviceBuffer(n * 32)
    buf.upload(vals)
    comm.sum_field_dev(buf.ptr, n)
    got = api.fr_to_canonical(buf.download((n, 4)))
    want = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(world * (world + 1) // 2)
    ok_field = bool(np.array_equal(got[:, 0], want) and not got[:, 1:].any())
    buf.free()
    comm.barrier()
    if rank == 0:
        print(json.dumps({"backend": dist.get_backend(), "world": world, "sum_disjoint": ok_disjoint, "gather_rows": bool(ok_rows), "sum_field_dev": ok_field}), flush=True)
    dist.destroy_process_group()
    assert ok_disjoint and ok_rows and ok_field


if __name__ == "__main__":
    main()
