#!/usr/bin/env python3
"""BASELINE configs[1] at full size — nearest_vector over 64 x 128-dim SIFT-shaped vectors, LOOKUP_BITS = 13, 2^14 rows, the result vector
public — proved by the GPU (ProverRounds) and by the oracle's CPU prover (oracle/prover.py) with the same blinding scalars: the proof
bytes and the verifying keys must be equal.  The same comparison as tests/test_gpu_cpu_prover.py makes at 2^11 – 2^13 rows, at a
BASELINE size (a minute or two of host time: run by hand, the result committed under profiles/).
usage (GPU box): python tools/c2_cpu_prover.py > gpurun_out/c2_cpu_prover.json"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api, circuit_sym as CS
from halo2_vectordb_amd.pipeline import NearestHotPath
from halo2_vectordb_amd.rounds import ProverRounds
from oracle import oracle as O, prover as PV

TAU, SEED = 0x1234567890ABCDEF1234567, 77
n, dim, k, P, L = 64, 128, 14, 48, 13
api.init(0)
t0 = time.time()
hp = NearestHotPath(n=n, dim=dim, k=k, P=P, L=L, tau=TAU).setup()
pr = ProverRounds(hp).keygen()
t_gpu_keygen = time.time() - t0
t0 = time.time()
got = pr.prove(None, seed=SEED)
api.sync()
t_gpu_prove = time.time() - t0
rows = O.quantize(hp.vectors_f64, P)
c = O.Ctx(store=True, keygen=True, plan_k=k)
c.assign_witnesses(rows[0])
c.assign_witnesses(rows[1:])
ind, res = c.nearest_vector("euclidean", rows[0], rows[1:], P=P, L=L)
cm, (_ind, res_cells) = CS.build_nearest("euclidean", n, dim, P, L, builder=None)
cs = PV.Circuit(k, L, c.break_points(), c.selectors(), c.n_lookup, cm.copy_of, cm.const_idx, cm.consts, cm.lookup_src, [int(x) for x in res_cells])
g, gl = O.srs_from_tau(k, TAU)
threads = min(16, len(os.sched_getaffinity(0)))
t0 = time.time()
pk = PV.keygen(cs, g, gl, threads=threads)
t_cpu_keygen = time.time() - t0
T = {}
t0 = time.time()
want = PV.prove(pk, c.advice(), c.lookup(), PV.seeded_blinds(cs, SEED), timings=T)
t_cpu_prove = time.time() - t0
vk_equal = all(np.array_equal(pk.commits[name], pr.fixed[name].commits) for name in PV.FIXED)
print(json.dumps({"circuit": "BASELINE C2: nearest_vector 64 x 128 + query, euclidean, P=48, LOOKUP_BITS=13, k=14, 128 public words",
                  "cells": len(c) + c.n_lookup, "advice_columns": cs.n_adv, "lookup_columns": cs.n_lk, "permutation_sets": cs.n_sets,
                  "proof_bytes": len(got["proof"]), "proof_sha256_gpu": hashlib.sha256(got["proof"]).hexdigest(),
                  "proof_sha256_cpu": hashlib.sha256(want["proof"]).hexdigest(), "proof_bytes_equal": got["proof"] == want["proof"],
                  "verifying_key_commitments_equal": bool(vk_equal), "instances_equal": got["instances"] == want["instances"],
                  "gpu_keygen_s": round(t_gpu_keygen, 2), "gpu_prove_ms": round(t_gpu_prove * 1e3, 1), "cpu_keygen_s": round(t_cpu_keygen, 1),
                  "cpu_prove_s": round(t_cpu_prove, 1), "cpu_stage_s": {a: round(b, 1) for a, b in T.items()}, "cpu_threads_for_commitments": threads}))
assert got["proof"] == want["proof"] and vk_equal
