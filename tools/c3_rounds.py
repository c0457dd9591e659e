import sys, time, json, numpy as np
sys.path.insert(0, ".")
from halo2_vectordb_amd import api
from halo2_vectordb_amd.pipeline import MerkleHotPath
from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
api.init(0)
t0 = time.time()
hp = MerkleHotPath(n=1024, dim=128, k=15).setup()
t1 = time.time()
pr = ProverRounds(hp).keygen()
t2 = time.time()
tied = int((pr.copy_of != np.arange(pr.copy_of.size)).sum())
T = {}
out = pr.prove(None, seed=1, timings=T)
T = {}
t3 = time.time()
out = pr.prove(None, seed=1, timings=T)
wall = (time.time() - t3) * 1e3
ok = quotient_identity_holds(pr, out["challenges"], out["evals"], out["instances"])
print(json.dumps({"workload": "merkle_commitment 1024x128 k=15 (BASELINE C3) with its full copy map and the constants gate", "cells": hp.n_cells, "columns": hp.n_cols,
                  "tied_cells": tied, "setup_s": round(t1 - t0, 1), "keygen_s": round(t2 - t1, 1), "quotient_identity_at_x_holds": bool(ok), "proof_bytes": len(out["proof"]),
                  "wall_ms": round(wall, 1), "device_ms": {k: round(v, 2) for k, v in T.items()}, "device_ms_total": round(sum(T.values()), 1)}))
