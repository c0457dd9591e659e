#!/usr/bin/env python3
"""merkle_commitment over n x dim vectors at 2^k rows as a whole proof on one MI355X: the Poseidon trace's constraint map (placed on the
device) in the permutation argument, the root public, fresh blinds, Fiat-Shamir transcript, SHPLONK.  Defaults: BASELINE C3
(1,024 x 128, k = 15); `--n 10000 --k 18 --block-cols 126 --ext-block-cols 64` is the database Merkle circuit of BASELINE C5
(1.54 G cells, 5,881 columns: 49 GB of columns, the cosets streamed).
`--circuit query` proves the reference's query circuit instead (examples/query.rs: nearest_vector + merkle_commitment over the same n
vectors, result vector and root public; `--lookup-bits`, `--metric`): e.g. the in-cluster query of BASELINE C5's demo,
`--circuit query --n 5000 --k 18 --lookup-bits 17 --block-cols 126 --ext-block-cols 64`."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo2_vectordb_amd import api  # noqa: E402
from halo2_vectordb_amd.pipeline import MerkleHotPath, QueryHotPath  # noqa: E402
from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--k", type=int, default=15)
ap.add_argument("--seed", type=int, default=20260003)
ap.add_argument("--block-cols", type=int, default=510)
ap.add_argument("--ext-block-cols", type=int, default=None)
ap.add_argument("--proofs", type=int, default=2)
ap.add_argument("--circuit", default="merkle", choices=["merkle", "query"])
ap.add_argument("--lookup-bits", type=int, default=13)
ap.add_argument("--metric", default="euclidean")
ap.add_argument("--out", default=None, help="write the proof (io.write_snark) and the verifying key beside it")
args = ap.parse_args()

api.init(0)
t0 = time.time()
if args.circuit == "query":
    hp = QueryHotPath(n=args.n, dim=args.dim, k=args.k, L=args.lookup_bits, metric=args.metric, seed=args.seed)
else:
    hp = MerkleHotPath(n=args.n, dim=args.dim, k=args.k, seed=args.seed)
hp.ext_block_cols = args.ext_block_cols
hp.setup()
t1 = time.time()
print(json.dumps({"setup_s": round(t1 - t0, 1), "cells": hp.n_cells, "columns": hp.n_cols, "free_GB": round(api.mem_info()[0] / 2**30, 1)}), file=sys.stderr, flush=True)
pr = ProverRounds(hp, block_cols=args.block_cols).keygen()
t2 = time.time()
print(json.dumps({"keygen_s": round(t2 - t1, 1), "mock": pr.keygen_report.as_dict(), "free_GB": round(api.mem_info()[0] / 2**30, 1)}), file=sys.stderr, flush=True)
best = None
for _ in range(args.proofs):
    t3 = time.time()
    out = pr.prove(None)
    api.sync()
    wall = (time.time() - t3) * 1e3
    print(json.dumps({"proof_wall_ms": round(wall, 1)}), file=sys.stderr, flush=True)
    if best is None or wall < best[0]:
        best = (wall, dict(pr.host_ms), out)
wall, host_ms, out = best
T = {}
pr.prove(None, timings=T)
ok = quotient_identity_holds(pr, out["challenges"], out["evals"], out["instances"])
root = api.fr_to_canonical(np.asarray(hp.results()[-1] if args.circuit == "query" else hp.results()).reshape(1, 4))[0]
root_int = int(root[0]) | int(root[1]) << 64 | int(root[2]) << 128 | int(root[3]) << 192
if args.out:
    from halo2_vectordb_amd.io import write_snark
    write_snark(args.out, out["proof"], out["instances"])
    pr.save_verifying_key(args.out + ".vk.npz", opened=out["opened"])
what = f"merkle_commitment {args.n}x{args.dim} k={args.k}" if args.circuit == "merkle" else \
    f"query circuit (nearest_vector {args.metric} + merkle_commitment) over {args.n}x{args.dim}, k={args.k}, LOOKUP_BITS={args.lookup_bits}"
print(json.dumps({"workload": what + ": whole constraint map, public outputs in the instance column, transcript, fresh blinds, SHPLONK", "lookup_cells": hp.n_lookup,
                  "cells": hp.n_cells, "columns": hp.n_cols, "product_sets": pr.n_sets, "mock_report_on_keygen_witness": pr.keygen_report.as_dict(),
                  "setup_s": round(t1 - t0, 1), "keygen_s": round(t2 - t1, 1), "quotient_identity_at_x_holds": bool(ok),
                  "public_root_is_the_hash_only_kernels_root": out["instances"][-1] == root_int, "n_instances": len(out["instances"]), "proof_bytes": len(out["proof"]), "proof_wall_ms": round(wall, 1),
                  "constraints_per_s": round(hp.n_cells / (wall * 1e-3)), "host_transcript_ms": round(host_ms["transcript"], 1),
                  "device_ms": {k: round(v, 2) for k, v in T.items()}, "device_ms_total": round(sum(T.values()), 1),
                  "block_cols": args.block_cols, "ext_cols_held": hp.ext_cols}))
pr.free()
hp.free()
