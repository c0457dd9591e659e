"""Exchange steps of the sharded prover (rounds.py, pipeline.py): one process per GPU, `torch.distributed` over RCCL
("nccl" on ROCm) on a multi-GPU node, gloo where ranks share a card or run on the CPU (tests).

The path shards by column, so what crosses between ranks is small and of two kinds only:

* values that exactly one rank produces and every rank needs in the global order — commitments (64 B per column), evaluations
  (32 B each), the end values of the running products, the handful of boundary polynomials: `sum_disjoint`, an integer
  all-reduce over arrays in which every slot is non-zero on at most one rank.  RCCL has no operator for field elements or curve
  points and needs none here: a sum with one non-zero contributor is exact.
* partial sums that every rank contributes to — the quotient's 2^(k+2) coefficients, SHPLONK's two partial commitments:
  `sum_field_dev` (all_gather of the partials, field additions on the device) and `gather_rows` (then `vdb_g1_sum`).
"""
import ctypes

import numpy as np


class LocalComm:
    """world of one: nothing to exchange"""
    rank, world = 0, 1
    exchange_ms = 0.0          # wall-clock ms spent inside exchange steps since the caller last reset it (Comm fills it)
    exchange_calls = 0

    def sum_disjoint(self, arr):
        return arr

    def gather_rows(self, row):
        return np.ascontiguousarray(row)[None]

    def sum_field_dev(self, ptr, n_elems):
        pass

    def barrier(self):
        pass


class Comm(LocalComm):
    """`dist`: the initialised torch.distributed module; the tensors of a collective live on the GPU with the nccl (RCCL)
    backend and on the host with gloo"""

    def __init__(self, dist):
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.on_gpu = dist.get_backend() == "nccl"
        self.device = torch.device("cuda", torch.cuda.current_device()) if self.on_gpu else torch.device("cpu")
        self.exchange_ms, self.exchange_calls = 0.0, 0

    def _timed(fn):
        """adds the call's wall time to exchange_ms: staging, the collective itself, and the wait for the slowest rank to arrive at it
        (a caller that wants a sharded proof's Amdahl terms resets the counters before the proof and reads them after)"""
        import functools
        import time

        @functools.wraps(fn)
        def wrapped(self, *a, **kw):
            t0 = time.perf_counter()
            try:
                return fn(self, *a, **kw)
            finally:
                self.exchange_ms += (time.perf_counter() - t0) * 1e3
                self.exchange_calls += 1
        return wrapped

    @_timed
    def sum_disjoint(self, arr):
        """arr: uint64 array of the same shape on every rank, every element non-zero on at most one rank -> the union"""
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        if a.size == 0:
            return a
        t = self.torch.from_numpy(a.view(np.int64).reshape(-1).copy()).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy().view(np.uint64).reshape(a.shape)

    def _gather_rows(self, row):
        """row: uint64 array of the same shape on every rank -> (world, ...) with every rank's"""
        a = np.ascontiguousarray(row, dtype=np.uint64)
        t = self.torch.from_numpy(a.view(np.int64).reshape(-1).copy()).to(self.device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.cpu().numpy().view(np.uint64).reshape(a.shape) for o in out])

    gather_rows = _timed(_gather_rows)

    def _accumulate_dev(self, ptr, everyone_ptr, n_elems):
        """ptr[i] += sum over r != rank of everyone[r][i] (mod r), on the device: `everyone_ptr` = world x n_elems field elements, rank
        r's partials at offset r x n_elems x 32 B.  The ONE accumulation loop of both backends (the RCCL branch hands in the gathered
        tensor, the gloo branch an upload of the gathered rows), so the world-2 gloo tests drive the offsets the RCCL run uses."""
        from . import api
        from ._lib import check
        lib = api.init()
        nbytes = int(n_elems) * 32
        one = np.array([0xAC96341C4FFFFFFB, 0x36FC76959F60CD29, 0x666EA36F7879462E, 0x0E0A77C19A07DF2F], dtype=np.uint64)   # R mod r: Montgomery one
        base = everyone_ptr.value if hasattr(everyone_ptr, "value") else int(everyone_ptr)
        for r in range(self.world):
            if r != self.rank:
                check(lib.vdb_poly_axpy_dev(ptr, api._p(one), ctypes.c_void_p(base + r * nbytes), ctypes.c_size_t(n_elems)))
        api.sync()          # the gathered buffer is its owner's to reuse from here on

    @_timed
    def sum_field_dev(self, ptr, n_elems):
        """the n_elems field elements at device pointer `ptr` <- their sum over all ranks (mod r), on every rank: all_gather of
        the partials (RCCL moves them GPU to GPU over xGMI; gloo stages them through the host) and world - 1 field additions per
        element on the device (_accumulate_dev).  The RCCL branch has run with world = 1 only so far (no multi-GPU node has been
        available: tests/test_gpu_sharded.py::test_comm_over_rccl_on_the_device); its collective is the one call the gloo tests
        do not cover."""
        from . import api
        from ._lib import check
        lib = api.init()
        nbytes = int(n_elems) * 32
        torch = self.torch
        if self.on_gpu:
            mine = torch.empty(n_elems * 4, dtype=torch.int64, device=self.device)
            check(lib.vdb_memcpy_d2d(ctypes.c_void_p(mine.data_ptr()), ptr, ctypes.c_size_t(nbytes)))
            api.sync()          # the library's stream has written `mine` before torch's stream reads it
            everyone = torch.empty(self.world * n_elems * 4, dtype=torch.int64, device=self.device)
            self.dist.all_gather_into_tensor(everyone, mine)
            torch.cuda.synchronize()      # ... and torch's stream has filled `everyone` before the library's stream reads it
            self._accumulate_dev(ptr, everyone.data_ptr(), n_elems)
            return
        host = np.empty((n_elems, 4), dtype=np.uint64)
        check(lib.vdb_memcpy_d2h(api._p(host), ptr, ctypes.c_size_t(nbytes)))
        parts = np.ascontiguousarray(self._gather_rows(host))
        everyone = api.DeviceBuffer(parts.nbytes)
        try:
            everyone.upload(parts)
            self._accumulate_dev(ptr, everyone.ptr, n_elems)
        finally:
            everyone.free()

    def barrier(self):
        self.dist.barrier()
