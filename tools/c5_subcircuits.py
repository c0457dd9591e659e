#!/usr/bin/env python3
"""BASELINE C5's two largest sub-circuits — the demo pipeline (tests/demo/mod.rs:47-90) on 10,000 x 128 vectors at 2^18 rows:
the database Merkle commitment (1.54 G cells, 5,876 columns, 49 GB of field elements) and kmeans<2, 1>
(0.95 G + 0.19 G cells, 3,615 + 713 columns) — through the hot path on ONE card, as the `world` column shards an 8-GPU job
would run side by side, here one after the other (each shard: witness cells of its columns -> commit -> transforms).
Prints per-shard and summed times; the Merkle root is checked against the hash-only kernel.
usage: c5_subcircuits.py [merkle|kmeans|nearest] [n] [world] [k]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api
from halo2_vectordb_amd._lib import check
from halo2_vectordb_amd.pipeline import KmeansHotPath, MerkleHotPath, NearestHotPath

which = sys.argv[1] if len(sys.argv) > 1 else "merkle"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
k = int(sys.argv[4]) if len(sys.argv) > 4 else 18
lib = api.init(0)
per_rank, root_ok, cols, cells = [], True, 0, 0
for rank in range(world):
    t0 = time.time()
    if which == "merkle":
        hp = MerkleHotPath(n=n, dim=128, k=k, seed=20260005, col_shard=(rank, world)).setup()
    elif which == "nearest":     # the query's nearest vector inside one cluster of ~n vectors
        hp = NearestHotPath(n=n, dim=128, k=k, L=k - 1, seed=20260005, col_shard=(rank, world)).setup()
    else:       # tests/demo/mod.rs:52: kmeans::<2, 1>; LOOKUP_BITS = k - 1
        hp = KmeansHotPath(n=n, dim=128, K=2, I=1, k=k, L=k - 1, seed=20260005, col_shard=(rank, world)).setup()
    t_setup = time.time() - t0
    hp.step()                                   # warm-up (twiddles, scratch)
    T = {}
    t0 = time.time()
    hp.step(T)
    wall = (time.time() - t0) * 1e3
    if which == "merkle":
        root_ok = root_ok and np.array_equal(hp.results(), api.poseidon_merkle_root(hp.qvec))
    per_rank.append({"rank": rank, "columns": hp.my_cols, "setup_s": round(t_setup, 1), "wall_ms": round(wall, 1), **{a: round(b, 2) for a, b in T.items()}})
    cols, cells = hp.n_cols, hp.n_cells + hp.n_lookup
    print(json.dumps(per_rank[-1]), file=sys.stderr, flush=True)
    hp.free()
    check(lib.vdb_scratch_release())            # the next shard's keygen-style setup needs the memory
tot = sum(r["wall_ms"] for r in per_rank)
name = {"merkle": f"merkle_commitment {n}x128 k={k} (BASELINE C5, database Merkle)", "kmeans": f"kmeans<2,1> {n}x128 k={k} L={k - 1} (BASELINE C5)",
        "nearest": f"nearest_vector in a cluster of {n}x128 k={k} L={k - 1} (BASELINE C5)"}[which]
print(json.dumps({"workload": name, "cells": cells, "columns": cols, "shards": world,
                  **({"root_matches_hash_only_kernel": bool(root_ok)} if which == "merkle" else {}), "sum_ms": round(tot, 1), "max_shard_ms": max(r["wall_ms"] for r in per_rank),
                  "constraints_per_s_one_card": cells / tot * 1e3, "per_shard": per_rank}))
