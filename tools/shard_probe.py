#!/usr/bin/env python3
"""Single-GPU emulation of one rank of an N-GPU job: per-stage times of rank r of w (no collectives)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api
from halo2_vectordb_amd.pipeline import KmeansHotPath

api.init(0)
import sys
worlds = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
for w in worlds:
    for r in (range(w) if len(worlds) == 1 else sorted({0, w // 2, w - 1})):
        hp = KmeansHotPath(col_shard=(r, w)).setup()
        hp.step()
        t = {}
        for _ in range(2):
            hp.step(t)
        import time
        t0 = time.perf_counter()
        for _ in range(5):
            hp.step()                      # as the bench's timed step without the collective: wall clock, host work included
        wall = (time.perf_counter() - t0) / 5 * 1e3
        print(json.dumps({"world": w, "rank": r, "my_cols": hp.my_cols, **{k: round(v / 2, 2) for k, v in t.items()}, "wall_ms_per_step": round(wall, 2)}), flush=True)
        hp.free()
