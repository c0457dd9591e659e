#!/usr/bin/env python3
"""SURVEY §8(d) micro-benchmarks on uniformly random field elements (the worst case for the MSM: every scalar is full width),
seed 1, 2^16 rows: commitment MSM with the default 11-bit window tables and with 14-bit tables, lagrange_to_coeff and
coeff_to_extended, per column and in aggregate.  (The realistic, skewed case — scalars from the generated witness — is
bench.py itself.)  usage: msm_ntt_micro.py [n_cols]"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api
from halo2_vectordb_amd._lib import check

lib = api.init(0)
n_cols, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 16
n = 1 << k
rng = np.random.default_rng(1)
raw = rng.integers(0, 1 << 62, size=(n_cols * n, 4), dtype=np.uint64)   # below r: valid Montgomery representatives of uniform elements
raw[:, 3] &= np.uint64((1 << 61) - 1)
d = api.DeviceBuffer(raw.nbytes)
d.upload(raw)
tau = np.array([12345, 0, 0, 0], dtype=np.uint64)
g, gl = api.srs_setup_unsafe(k, api.fr_from_canonical(tau.reshape(1, 4))[0])
out = {"n_cols": n_cols, "rows": n, "scalars": "uniform Fr, seed 1"}
pts = np.zeros((n_cols, 8), dtype=np.uint64)
for bits in (0, 14):
    srs = api.Srs(k, None, gl, window_bits=bits)
    for it in range(3):
        api.timer_start()
        check(lib.vdb_msm_batch_dev(srs.h, 1, d.ptr, ctypes.c_size_t(n_cols), ctypes.c_size_t(n), api._p(pts)))
        ms = api.timer_stop()
    _, c, w = srs.info()
    out[f"msm_c{c}"] = {"windows": w, "ms": round(ms, 2), "us_per_column": round(ms / n_cols * 1e3, 1), "points_per_s": n_cols * n / ms * 1e3,
                        "bucket_adds_per_s": n_cols * n * w / ms * 1e3}
    srs.free()
ext = api.DeviceBuffer(n_cols * n * 4 * 32)
for it in range(3):
    api.timer_start()
    check(lib.vdb_lagrange_to_coeff_dev(d.ptr, ctypes.c_size_t(n_cols), k))
    ms_i = api.timer_stop()
    api.timer_start()
    check(lib.vdb_coeff_to_extended_dev(d.ptr, ext.ptr, ctypes.c_size_t(n_cols), k, 2))
    ms_e = api.timer_stop()
out["lagrange_to_coeff"] = {"ms": round(ms_i, 2), "us_per_column": round(ms_i / n_cols * 1e3, 1), "butterflies_per_s": n_cols * (n // 2) * k / ms_i * 1e3}
out["coeff_to_extended_4n"] = {"ms": round(ms_e, 2), "us_per_column": round(ms_e / n_cols * 1e3, 1), "butterflies_per_s": n_cols * (2 * n) * k / ms_e * 1e3}
print(json.dumps(out))
