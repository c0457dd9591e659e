#!/usr/bin/env python3
"""Per-kernel HIP-event times of one step of the C4 hot path (all kernels, not just the top 8 bench.py prints)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api
from halo2_vectordb_amd.pipeline import KmeansHotPath

api.init(0)
r, w = (int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (0, 1)))
hp = KmeansHotPath(col_shard=(r, w)).setup()
hp.step()
api.profile_begin()
hp.step()
prof = api.profile_end()
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:28s} {v['ms']:9.3f} ms  {v['launches']:5d} launches  {v['ms'] / v['launches'] * 1e3:9.1f} us/launch")
print("total", sum(v["ms"] for v in prof.values()))
