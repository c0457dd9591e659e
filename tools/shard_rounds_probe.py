#!/usr/bin/env python3
"""Single-GPU EMULATION of one rank of an N-GPU sharded proof (rounds.ProverRounds with world > 1): rank r of w runs its own share of the
rounds with a stand-in communicator that returns what the exchanges would return with every other rank's contribution left out (zero
points, ones for scalars) — the proof bytes are meaningless, the work and its timing are the rank's: per-stage device times and the
wall time of an untimed proof (host transcript included, which every rank replicates).  NOT a multi-GPU measurement: no exchange is
timed, ranks run one at a time.  usage: shard_rounds_probe.py [world [rank ...]]   (default: world 8, ranks 0, 3, 7)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api  # noqa: E402
from halo2_vectordb_amd.pipeline import KmeansHotPath  # noqa: E402
from halo2_vectordb_amd.rounds import ProverRounds  # noqa: E402

ONE = np.array([0xAC96341C4FFFFFFB, 0x36FC76959F60CD29, 0x666EA36F7879462E, 0x0E0A77C19A07DF2F], dtype=np.uint64)   # Montgomery one


class AloneComm:
    """what a rank sees of the exchanges when nobody else contributes"""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def sum_disjoint(self, arr):
        a = np.array(arr, dtype=np.uint64, copy=True)
        if a.ndim == 2 and a.shape[1] == 4:          # scalars: a zero would wipe out running products and make later MSMs trivial
            a[~a.any(axis=1)] = ONE
        return a

    def gather_rows(self, row):
        return np.stack([np.ascontiguousarray(row, dtype=np.uint64)] * self.world)

    def sum_field_dev(self, ptr, n_elems):
        pass

    def barrier(self):
        pass


api.init(0)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(x) for x in sys.argv[2:]] or sorted({0, world // 2 - 1, world - 1})
I = int(os.environ.get("VDB_PROBE_I", "8"))
for r in ranks:
    t0 = time.time()
    hp = KmeansHotPath(n=256, dim=128, K=4, I=I, k=16, P=48, L=15, metric="cosine", col_shard=(r, world))
    hp.ext_block_cols = 256
    hp.setup()
    pr = ProverRounds(hp, comm=AloneComm(r, world)).keygen(check=False)
    t_key = time.time() - t0
    pr.prove(None)
    walls = []
    for _ in range(2):
        t0 = time.perf_counter()
        pr.prove(None)
        api.sync()
        walls.append((time.perf_counter() - t0) * 1e3)
    host = dict(pr.host_ms)
    T = {}
    pr.prove(None, timings=T)
    print(json.dumps({"world": world, "rank": r, "my_advice_cols": hp.my_adv, "my_lookup_cols": hp.my_lk, "my_sets": pr.my_sets, "foreign": pr.foreign, "stray": pr.stray,
                      "setup_keygen_s": round(t_key, 1), "proof_wall_ms": round(min(walls), 1), "host_transcript_ms": round(host["transcript"], 1),
                      "device_ms": {k: round(v, 1) for k, v in T.items()}, "device_ms_total": round(sum(T.values()), 1)}), flush=True)
    pr.free()
    hp.free()
