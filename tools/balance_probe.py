#!/usr/bin/env python3
"""Per-rank MSM entry totals against measured kernel times (ranks emulated one after another on one GPU)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api
from halo2_vectordb_amd.pipeline import KmeansHotPath

api.init(0)
w = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for r in range(w):
    hp = KmeansHotPath(col_shard=(r, w)).setup()
    ea, el = hp.msm_entries
    mine = int(ea[hp.a_lo:hp.a_hi].sum() + el[hp.l_lo:hp.l_hi].sum())
    hp.step()
    api.profile_begin()
    hp.step()
    prof = api.profile_end()
    t = {k: round(v["ms"], 2) for k, v in prof.items() if k.startswith("k_msm") or k == "k_ntt_pass"}
    print(json.dumps({"rank": r, "adv": [hp.a_lo, hp.a_hi], "lk": [hp.l_lo, hp.l_hi], "cols": hp.my_cols, "entries_M": round(mine / 1e6, 1), **t}), flush=True)
    hp.free()
