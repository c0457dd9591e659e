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
# gate numerator on the extended coset: 512 columns of 2^18 (any field elements do for timing)
gc, k_, e_ = 512, 16, 2
ne = 1 << (k_ + e_)
d_adv, d_sel, d_acc = api.DeviceBuffer(gc * ne * 32), api.DeviceBuffer(gc * ne * 32), api.DeviceBuffer(ne * 32)
for c0 in range(0, gc * ne * 32, raw.nbytes):
    nb = min(raw.nbytes, gc * ne * 32 - c0)
    check(lib.vdb_memcpy_d2d(ctypes.c_void_p(d_adv.ptr.value + c0), bufs[0].ptr, ctypes.c_size_t(nb)))
    check(lib.vdb_memcpy_d2d(ctypes.c_void_p(d_sel.ptr.value + c0), bufs[1].ptr, ctypes.c_size_t(nb)))
check(lib.vdb_memset_dev(d_acc.ptr, 0, ne * 32))
for it in range(3):
    api.timer_start()
    check(lib.vdb_gate_eval_dev(d_adv.ptr, d_sel.ptr, ctypes.c_size_t(gc), k_, e_, api._p(x), d_acc.ptr))
    ms_g = api.timer_stop()
print(json.dumps({"kernel": "k_gate_eval", "n_cols": gc, "ext_rows": ne, "ms": round(ms_g, 3), "cells_per_s": gc * ne / ms_g * 1e3,
                  "algorithmic_GBps": 64.0 * gc * ne / ms_g / 1e6, "fr_mul_per_s": 3.0 * gc * ne / ms_g * 1e3}))
for b in (d_adv, d_sel, d_acc):
    b.free()
# lookup argument: C4's 1,345 lookup columns of 15-bit cells against the range table
lk_cols, bits, usable = 1345, 15, n - 6
vals = rng.integers(0, 1 << bits, size=lk_cols * n, dtype=np.uint64)
R_MONT = (1 << 256) % 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
mont = lambda arr: api.fr_from_canonical(np.stack([arr, np.zeros_like(arr), np.zeros_like(arr), np.zeros_like(arr)], axis=1))
d_in = api.DeviceBuffer(lk_cols * n * 32)
for c0 in range(0, lk_cols, 128):
    c1 = min(lk_cols, c0 + 128)
    d_in.upload(mont(vals[c0 * n: c1 * n]), offset=c0 * n * 32)
tab = np.arange(n, dtype=np.uint64)
tab[tab >= (1 << bits)] = 0
d_tab = api.DeviceBuffer(n * 32)
d_tab.upload(mont(tab))
d_oa, d_os = api.DeviceBuffer(lk_cols * n * 32), api.DeviceBuffer(lk_cols * n * 32)
for it in range(3):
    api.timer_start()
    check(lib.vdb_lookup_permute_dev(d_in.ptr, d_tab.ptr, ctypes.c_size_t(lk_cols), ctypes.c_size_t(n), ctypes.c_size_t(usable), bits, d_oa.ptr, d_os.ptr))
    ms_lp = api.timer_stop()
print(json.dumps({"kernel": "k_lp_hist + k_lp_scan + k_lp_rows", "n_cols": lk_cols, "n": n, "bits": bits, "ms": round(ms_lp, 3),
                  "rows_per_s": lk_cols * n / ms_lp * 1e3, "algorithmic_GBps": 96.0 * lk_cols * n / ms_lp / 1e6}))
# lookup product over the same columns (terms + grand product), then the permutation product of 2,048 advice columns
beta, gamma = x, x[::-1].copy()
d_z = api.DeviceBuffer(lk_cols * n * 32)
for it in range(3):
    api.timer_start()
    check(lib.vdb_lookup_product_dev(d_in.ptr, d_tab.ptr, d_oa.ptr, d_os.ptr, ctypes.c_size_t(lk_cols), ctypes.c_size_t(n), ctypes.c_size_t(usable),
                                     api._p(beta), api._p(gamma), d_z.ptr))
    ms_lz = api.timer_stop()
print(json.dumps({"kernel": "k_lookup_terms + k_grand_product", "n_cols": lk_cols, "n": n, "ms": round(ms_lz, 3), "rows_per_s": lk_cols * n / ms_lz * 1e3}))
for b in (d_in, d_oa, d_os, d_z):
    b.free()
chunk = 3
n_chunks = -(-n_cols // chunk)
d_pz = api.DeviceBuffer(n_chunks * n * 32)
delta = api.fr_delta()
for it in range(3):
    api.timer_start()
    check(lib.vdb_permutation_product_dev(bufs[0].ptr, bufs[1].ptr, ctypes.c_size_t(n_cols), 16, ctypes.c_size_t(usable), ctypes.c_size_t(chunk),
                                          api._p(beta), api._p(gamma), api._p(delta), d_pz.ptr))
    ms_pz = api.timer_stop()
print(json.dumps({"kernel": "k_perm_terms + k_grand_product + chain", "n_cols": n_cols, "chunk_len": chunk, "n": n, "ms": round(ms_pz, 3),
                  "cells_per_s": n_cols * n / ms_pz * 1e3, "algorithmic_GBps": (64.0 * n_cols + 32.0 * n_chunks) * n / ms_pz / 1e6}))
print(json.dumps({"kernel": "k_grand_product", "n_cols": n_cols, "n": n, "ms": round(ms, 3), "rows_per_s": rows / ms * 1e3, "algorithmic_GBps": 224.0 * rows / ms / 1e6,
                  "fr_mul_per_s": 5.0 * rows / ms * 1e3}))
