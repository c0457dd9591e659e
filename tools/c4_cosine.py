#!/usr/bin/env python3
"""BASELINE C4' — the satisfiable variant of the k = 16 k-means circuit: kmeans::<4, 8> over 256 x 128 vectors with the COSINE
distance, what the reference's example actually runs (examples/kmeans.rs:48-49; SURVEY §8d: 1.09 G advice + 0.24 G lookup
cells, 20.3 k columns, 42.6 GB of field elements, 170 GB of extended cosets) — through the hot path on ONE card: the
streams, the coefficient columns and the MSM work space stay resident, the extended cosets are produced in column blocks
(pipeline.KmeansHotPath.ext_cols).  Prints ms / step and constraints / s beside the Euclidean line of bench.py.
usage: c4_cosine.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api
from halo2_vectordb_amd.pipeline import KmeansHotPath

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
api.init(0)
t0 = time.time()
hp = KmeansHotPath(n=256, dim=128, K=4, I=8, k=16, P=48, L=15, metric="cosine").setup()
setup_s = time.time() - t0
print(json.dumps({"setup_s": round(setup_s, 1), "cells": hp.n_cells, "lookups": hp.n_lookup, "columns": hp.n_cols, "ext_cols": hp.ext_cols}), file=sys.stderr, flush=True)
hp.step()
T = {}
api.profile_begin(deferred=True)
t0 = time.perf_counter()
for _ in range(steps):
    hp.step(T)
api.sync()
el = time.perf_counter() - t0
prof = api.profile_end()
cells = hp.n_cells + hp.n_lookup
free, total = api.mem_info()
print(json.dumps({
    "workload": "kmeans K=4 I=8 over 256x128 SIFT-shaped vectors, P=48, LOOKUP_BITS=15, COSINE, k=16 (BASELINE C4', the satisfiable variant)",
    "advice_cells": hp.n_cells, "lookup_cells": hp.n_lookup, "advice_columns": hp.n_adv_cols, "lookup_columns": hp.n_lk_cols,
    "ext_columns_resident": hp.ext_cols, "ms_per_step": el / steps * 1e3, "constraints_per_s": cells * steps / el,
    "stage_ms": {k: v / steps for k, v in T.items()}, "hbm_used_gb": round((total - free) / 1e9, 1),
    "kernels_ms_per_step": {k: round(v["ms"] / steps, 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:8]},
    "setup_s": round(setup_s, 1)}))
hp.free()
