#!/usr/bin/env python3
"""One proof with fixed blinding seeds, on one rank or sharded over the ranks of a torch.distributed launch — the proof bytes
must not depend on the number of ranks (tests/test_gpu_sharded.py).

    python tools/sharded_prove.py --circuit kmeans --out /tmp/p1.bin
    VDB_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
        tools/sharded_prove.py --circuit kmeans --out /tmp/p2.bin

Circuits: "kmeans" (a small cosine k-means), "nearest" (small), "c2" (BASELINE configs[1]: nearest_vector over 64 x 128, k = 14),
"merkle" (small, no lookup columns), "distances" (examples/distances.rs, small), "query" (small: nearest_vector and merkle_commitment in one circuit), "mid" (cosine k-means over 128 x 64 vectors, K = 4, I = 4, at 2^16 rows).  Rank 0 writes the proof bytes to --out and prints one JSON line; every rank checks that its
own transcript ended with the same bytes (sha256 exchanged)."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TAU = 0x1234567890ABCDEF1234567


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--circuit", default="kmeans")
    ap.add_argument("--out", default=None)
    ap.add_argument("--seed", type=int, default=41)
    ap.add_argument("--block-cols", type=int, default=510)
    ap.add_argument("--ext-block-cols", type=int, default=None, help="hold this many coset columns (forces the streamed rounds)")
    ap.add_argument("--timed", type=int, default=0, help="also time this many proofs with fresh blinds")
    ap.add_argument("--save-key", default=None, help="after keygen every rank writes its share of the proving key to <path>.rank<r>of<w>.npz")
    ap.add_argument("--load-key", default=None, help="no keygen: every rank reads its share of the proving key from <path>.rank<r>of<w>.npz")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    comm = None
    from halo2_vectordb_amd import api
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VDB_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dev = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(dev)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
            api.init(dev)
        else:
            dist.init_process_group(backend=backend)
            api.init(0)
        from halo2_vectordb_amd.dist import Comm
        comm = Comm(dist)
    else:
        api.init(0)
    from halo2_vectordb_amd.pipeline import KmeansHotPath, MerkleHotPath, NearestHotPath, QueryHotPath
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    shard = (rank, world)
    if args.circuit.startswith("kmeans:"):    # a shape of the caller's: kmeans:n,dim,K,I,k,L,metric (the tests' seeded random shapes)
        n_, dim_, K_, I_, k_, L_, metric_ = args.circuit.split(":", 1)[1].split(",")
        hp = KmeansHotPath(n=int(n_), dim=int(dim_), K=int(K_), I=int(I_), k=int(k_), L=int(L_), metric=metric_, tau=TAU, col_shard=shard)
    elif args.circuit == "kmeans":
        hp = KmeansHotPath(n=8, dim=4, K=2, I=2, k=12, L=11, metric="cosine", tau=TAU, col_shard=shard)
    elif args.circuit == "nearest":
        hp = NearestHotPath(n=6, dim=4, k=12, L=11, tau=TAU, col_shard=shard)
    elif args.circuit == "query":    # examples/query.rs: nearest_vector + merkle_commitment in one circuit, result vector and root public
        hp = QueryHotPath(n=6, dim=4, k=12, L=11, metric="cosine", tau=TAU, col_shard=shard)
    elif args.circuit == "c2":
        hp = NearestHotPath(n=64, dim=128, k=14, L=13, tau=TAU, col_shard=shard)
    elif args.circuit == "mid":      # a k = 16 cosine k-means of a few thousand columns: per-rank work and exchange overheads at a real row count
        hp = KmeansHotPath(n=128, dim=64, K=4, I=4, k=16, P=48, L=15, metric="cosine", tau=TAU, col_shard=shard)
    elif args.circuit == "half":     # BASELINE C4' with half the iterations: 0.69 G cells, ~10.5 k columns at 2^16 rows (two ranks fit one 288 GB card)
        hp = KmeansHotPath(n=256, dim=128, K=4, I=4, k=16, P=48, L=15, metric="cosine", tau=TAU, col_shard=shard)
    elif args.circuit == "merkle":
        hp = MerkleHotPath(n=6, dim=5, k=11, tau=TAU, col_shard=shard)
    elif args.circuit == "distances":   # examples/distances.rs: three distances of two vectors, each public (a dozen columns at 2^10 rows)
        from halo2_vectordb_amd.pipeline import DistancesHotPath
        hp = DistancesHotPath(dim=6, k=10, L=9, tau=TAU, col_shard=shard)
    else:
        raise SystemExit("unknown circuit")
    hp.ext_block_cols = args.ext_block_cols
    t0 = time.perf_counter()
    hp.setup()
    key_file = lambda base: f"{base}.rank{rank}of{world}.npz"
    if args.load_key:
        pr = ProverRounds(hp, block_cols=args.block_cols, comm=comm).load_proving_key(key_file(args.load_key))
    else:
        pr = ProverRounds(hp, block_cols=args.block_cols, comm=comm).keygen()
    keygen_s = time.perf_counter() - t0
    if args.save_key:
        pr.save_proving_key(key_file(args.save_key))
    out = pr.prove(None, seed=args.seed)
    proof = out["proof"]
    digest = hashlib.sha256(proof).digest()
    same = True
    if comm is not None:
        rows = comm.gather_rows(np.frombuffer(digest, dtype=np.uint64))
        same = bool((rows == rows[0]).all())
    ok = bool(quotient_identity_holds(pr, out["challenges"], out["evals"], out["instances"]))
    wall = []
    for _ in range(args.timed):
        if comm is not None:
            api.sync()
            comm.barrier()
        t0 = time.perf_counter()
        pr.prove(None)
        api.sync()
        wall.append((time.perf_counter() - t0) * 1e3)
    if rank == 0:
        if args.out:
            with open(args.out, "wb") as f:
                f.write(proof)
            np.savez(args.out + ".vk.npz", **{"fixed_" + name: pr.fixed[name].commits for name in pr.fixed}, instances=np.array([str(v) for v in out["instances"]]),
                     meta=np.frombuffer(json.dumps(dict(rows=pr.rows, k=pr.k, n_adv=pr.n_adv, n_lk=pr.n_lk, n_cols=pr.n_cols, n_sets=pr.n_sets, chunk_len=pr.chunk_len,
                                                        opened={str(r): v for r, v in out["opened"].items()})).encode(), dtype=np.uint8))
        print(json.dumps({"circuit": args.circuit, "world": world, "columns": pr.n_cols, "sets": pr.n_sets, "proof_bytes": len(proof),
                          "sha256": digest.hex(), "every_rank_wrote_the_same_bytes": same, "quotient_identity_at_x_holds": ok,
                          "mock_prover_violations": pr.keygen_report.violations() if hasattr(pr, "keygen_report") else None, "key_loaded_from_file": bool(args.load_key),
                          "n_instances": len(out["instances"]),
                          "shards": [[list(a), list(l)] for a, l in hp.shards], "my_set_ranges": pr.set_ranges, "foreign": pr.foreign, "stray": pr.stray,
                          "keygen_s": round(keygen_s, 2), "proof_ms": [round(w, 1) for w in wall]}), flush=True)
    assert same and ok
    pr.free()
    hp.free()
    if comm is not None:
        comm.barrier()
        comm.dist.destroy_process_group()


if __name__ == "__main__":
    main()
