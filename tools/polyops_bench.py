#!/usr/bin/env python3
"""Throughput of vdb_grand_product_dev and vdb_eval_polys_dev on device-resident columns (SURVEY §8 f1 bricks)."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api
from halo2_vectordb_amd._lib import check

lib = api.init(0)
n_cols, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048), 1 << 16
rng = np.random.default_rng(1)
# any non-zero 255-bit patterns below r are valid Montgomery representatives
raw = rng.integers(1, 1 << 62, size=(n_cols * n, 4), dtype=np.uint64)
raw[:, 3] &= (1 << 60) - 1
bufs = [api.DeviceBuffer(raw.nbytes) for _ in range(3)]
bufs[0].upload(raw)
bufs[1].upload(raw[::-1].copy())
for it in range(3):
    api.timer_start()
    check(lib.vdb_grand_product_dev(bufs[0].ptr, bufs[1].ptr, ctypes.c_size_t(n_cols), ctypes.c_size_t(n), bufs[2].ptr))
    ms = api.timer_stop()
rows = n_cols * n
x = np.array([0x1234567890ABCDEF, 0x0FEDCBA098765432, 0x1111111122222222, 0x0333333344444444], dtype=np.uint64)
out = np.zeros((n_cols, 4), dtype=np.uint64)
for it in range(3):
    api.timer_start()
    check(lib.vdb_eval_polys_dev(bufs[0].ptr, ctypes.c_size_t(n_cols), ctypes.c_size_t(n), api._p(x), api._p(out)))
    ms_ev = api.timer_stop()
print(json.dumps({"kernel": "k_eval_polys", "n_cols": n_cols, "n": n, "ms": round(ms_ev, 3), "coeff_per_s": rows / ms_ev * 1e3,
                  "algorithmic_GBps": 32.0 * rows / ms_ev / 1e6, "hbm_frac": 32.0 * rows / ms_ev / 1e6 / 8000.0}))
print(json.dumps({"kernel": "k_grand_product", "n_cols": n_cols, "n": n, "ms": round(ms, 3), "rows_per_s": rows / ms * 1e3, "algorithmic_GBps": 224.0 * rows / ms / 1e6,
                  "fr_mul_per_s": 5.0 * rows / ms * 1e3}))
