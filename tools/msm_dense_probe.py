#!/usr/bin/env python3
"""k_msm_sort / k_msm_accum on dense (uniformly random) scalars with the 14-bit windows the product columns use: per-column kernel
times from the library's HIP-event profile.  Round 3 used it for an A/B of a sort that kept every long scalar's signed digits between
its two visits (one Montgomery reduction and one digit extraction per scalar instead of three and two): 15.4 against 15.8 us per column —
the dense sort is bound by its 2.5 M LDS atomics and 1.2 M scattered 4-byte entry stores per column, not by the field work; not kept
(profiles/r03_msm_dense_probe.json)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api  # noqa: E402

api.init(0)
k, n_cols = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = 1 << k
rng = np.random.default_rng(1)
tau = np.array([5, 0, 0, 0], dtype=np.uint64)
g, gl = api.srs_setup_unsafe(k, api.fr_from_canonical(tau.reshape(1, 4))[0])
srs = api.Srs(k, None, gl, window_bits=14)
buf = api.DeviceBuffer(n_cols * n * 32)
api.random_scalars_dev(buf.ptr, n_cols * n, seed=3)
api.msm_batch_dev(srs, buf.ptr, n_cols, n)
import time
api.sync()
t0 = time.perf_counter()
for _ in range(3):
    out = api.msm_batch_dev(srs, buf.ptr, n_cols, n)
api.sync()
wall_us = (time.perf_counter() - t0) * 1e6 / 3 / n_cols      # sorts and accumulations overlap here (VDB_MSM_PIPELINE=0: they do not)
api.profile_begin()
for _ in range(3):
    out = api.msm_batch_dev(srs, buf.ptr, n_cols, n)
prof = api.profile_end()
print(json.dumps({"columns": n_cols, "k": k, "window_bits": 14, "wall_us_per_column": round(wall_us, 2), "pipeline": os.environ.get("VDB_MSM_PIPELINE", "1"),                   "us_per_column": {name: round(rec["ms"] * 1e3 / 3 / n_cols, 2) for name, rec in prof.items()},
                  "checksum": int(np.bitwise_xor.reduce(out.reshape(-1)))}))
