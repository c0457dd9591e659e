#!/usr/bin/env python3
"""Stage times of the prover rounds (halo2_vectordb_amd/rounds.py) on a k-means circuit at 2^16 rows.
usage: rounds_bench.py [I]   — I k-means iterations of the BASELINE C4 shape (K=4, 256 x 128)
(I = 8 is the BASELINE C4 circuit; the default I = 2 runs in a fraction of the memory and time)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api
from halo2_vectordb_amd.pipeline import KmeansHotPath
from halo2_vectordb_amd.rounds import ProverRounds

I = int(sys.argv[1]) if len(sys.argv) > 1 else 2
api.init(0)
t0 = time.time()
hp = KmeansHotPath(I=I).setup()
pr = ProverRounds(hp).keygen()
t_setup = time.time() - t0
rng = np.random.default_rng(1)
raw = rng.integers(1, 1 << 62, size=(7, 4), dtype=np.uint64)
raw[:, 3] &= np.uint64((1 << 60) - 1)
ch = dict(zip(("beta", "gamma", "y", "x", "v", "yo", "u"), raw))
best = None
for it in range(2):
    T = {}
    t0 = time.time()
    out = pr.prove(ch, seed=it, timings=T)
    wall = (time.time() - t0) * 1e3
    if best is None or wall < best[0]:
        best = (wall, T)
wall, T = best
from halo2_vectordb_amd.rounds import quotient_identity_holds
identity = quotient_identity_holds(pr, out["challenges"], out["evals"], out["instances"])
cells = hp.n_cells + hp.n_lookup
print(json.dumps({"workload": f"kmeans K=4 I={I} 256x128 k=16", "advice_columns": pr.n_adv, "lookup_columns": pr.n_lk, "product_columns": pr.n_sets + pr.n_lk,
                  "cells": cells, "quotient_identity_at_x_holds": bool(identity), "setup_s": round(t_setup, 1), "wall_ms": round(wall, 1), "device_ms": {k: round(v, 2) for k, v in T.items()},
                  "device_ms_total": round(sum(T.values()), 1)}))
pr.free()
hp.free()
