"""The prover rounds sharded over ranks (halo2_vectordb_amd/rounds.py with a dist.Comm; SURVEY §8e, north_star's "partitioned
across the GPUs of one node ... all-reduce only for the final coefficient reduction"): a proof made by two or three ranks —
launched as the driver launches bench.py, torch.distributed.run with one process per rank, here sharing this box's one GPU over
gloo (VDB_DIST_BACKEND=gloo; the same code runs over RCCL on a multi-GPU node) — is, with the same blinding seeds, BYTE FOR BYTE
the proof one rank makes; and the stand-alone verifier accepts those bytes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, circuit, out, port, extra=()):
    env = dict(os.environ, VDB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    script = os.path.join(ROOT, "tools", "sharded_prove.py")
    if world == 1:
        cmd = [sys.executable, script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port), script]
    res = subprocess.run(cmd + ["--circuit", circuit, "--out", out, *extra], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    return json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("circuit,worlds", [("kmeans", (2, 3, 4)), ("merkle", (2, 4)), ("query", (2, 3)), ("c2", (2,)), ("distances", (2,))])
def test_sharded_proof_is_the_single_rank_proof(tmp_path, circuit, worlds):
    one = _run(1, circuit, str(tmp_path / "p1.bin"), 0)
    assert one["every_rank_wrote_the_same_bytes"] and one["quotient_identity_at_x_holds"] and one["mock_prover_violations"] == 0
    want = open(tmp_path / "p1.bin", "rb").read()
    for i, world in enumerate(worlds):
        rep = _run(world, circuit, str(tmp_path / f"p{world}.bin"), 29541 + i)
        assert rep["world"] == world and rep["every_rank_wrote_the_same_bytes"] and rep["quotient_identity_at_x_holds"]
        got = open(tmp_path / f"p{world}.bin", "rb").read()
        assert got == want, (circuit, world, rep)
        assert rep["sha256"] == one["sha256"] and rep["n_instances"] == one["n_instances"]


def test_sharded_proofs_of_random_kmeans_shapes(tmp_path):
    """seeded random k-means circuits (vectors, width, clusters, iterations, column height, lookup width, cosine or manhattan) on three and
    four ranks: where the column blocks, the sets of the permutation argument and the advice / lookup junction fall is different every
    time — the bytes are the one-rank proof's"""
    rng = np.random.default_rng(2718)
    for case in range(3):
        n, dim = int(rng.integers(5, 12)), int(rng.integers(2, 7))
        K, I = int(rng.integers(1, 4)), int(rng.integers(1, 3))
        k = int(rng.integers(10, 13))
        L = int(rng.integers(8, k))
        metric = ("cosine", "manhattan")[case % 2]
        circuit = f"kmeans:{n},{dim},{K},{I},{k},{L},{metric}"
        one = _run(1, circuit, str(tmp_path / f"r{case}_1.bin"), 29560 + 3 * case)
        want = open(tmp_path / f"r{case}_1.bin", "rb").read()
        assert one["quotient_identity_at_x_holds"] and len(want) > 0
        world = 3 + case % 2
        rep = _run(world, circuit, str(tmp_path / f"r{case}_{world}.bin"), 29561 + 3 * case)
        assert rep["world"] == world and rep["every_rank_wrote_the_same_bytes"] and rep["quotient_identity_at_x_holds"], (circuit, rep)
        assert open(tmp_path / f"r{case}_{world}.bin", "rb").read() == want, (circuit, world)


def test_sharded_streamed_proof_and_the_verifier(tmp_path, O):
    """two ranks, cosets held 7 columns at a time and blocks of 12 columns (the streamed rounds): the same bytes again, and the
    stand-alone verifier of tests/test_gpu_rounds.py accepts them against the gathered fixed commitments"""
    from halo2_vectordb_amd import api
    from oracle import pairing as PR
    from test_gpu_rounds import TAU, _verify
    api.init(0)
    one = _run(1, "kmeans", str(tmp_path / "p1.bin"), 0)
    two = _run(2, "kmeans", str(tmp_path / "p2.bin"), 29547, extra=("--ext-block-cols", "7", "--block-cols", "12"))
    proof = open(tmp_path / "p2.bin", "rb").read()
    assert proof == open(tmp_path / "p1.bin", "rb").read()
    with np.load(str(tmp_path / "p2.bin") + ".vk.npz", allow_pickle=False) as doc:
        meta = json.loads(bytes(doc["meta"]).decode())
        fixed = {name[6:]: np.ascontiguousarray(doc[name]) for name in doc.files if name.startswith("fixed_")}
        instances = [int(v) for v in doc["instances"]]
    with np.load(str(tmp_path / "p1.bin") + ".vk.npz", allow_pickle=False) as doc:
        for name in fixed:                       # the sharded keygen gathers the same verifying key
            assert np.array_equal(fixed[name], doc["fixed_" + name]), name
    from halo2_vectordb_amd.rounds import N_BLIND, _fr_to_int
    vk = dict(meta=dict(rows=meta["rows"], k=meta["k"], n_adv=meta["n_adv"], n_lk=meta["n_lk"], n_cols=meta["n_cols"], n_sets=meta["n_sets"], chunk_len=meta["chunk_len"],
                        n_blind=N_BLIND, delta=_fr_to_int(api.fr_delta()), n_instances=len(instances)),
              opened={int(r): v for r, v in meta["opened"].items()}, fixed=fixed, tau_h=PR.pt_mul(PR.G2, TAU), instances=instances)
    assert _verify(O, api, proof, vk)
    assert not _verify(O, api, proof, {**vk, "instances": instances[:-1] + [(instances[-1] + 1) % O.R_MOD]})
    assert one["sha256"] == two["sha256"]


@pytest.mark.parametrize("circuit", ["kmeans", "merkle"])
def test_sharded_proving_key_round_trip(tmp_path, circuit):
    """The Keygen arm's files for a sharded deployment (src/scaffold/mod.rs:267-283 writes pk + pinning once, :290-291 reads them for
    every proof): two ranks keygen and each writes its share of the proving key (ProverRounds.save_proving_key: its selectors, the
    sigma columns of its sets, the whole set's commitments); a second launch loads the shares instead of running keygen and writes the
    same proof bytes — which are the one-rank proof's.  A share is refused by another rank and by a run with another partition."""
    one = _run(1, circuit, str(tmp_path / "p1.bin"), 0)
    key = str(tmp_path / "key")
    made = _run(2, circuit, str(tmp_path / "p2a.bin"), 29551, extra=("--save-key", key))
    for r in (0, 1):
        assert os.path.exists(f"{key}.rank{r}of2.npz")
    loaded = _run(2, circuit, str(tmp_path / "p2b.bin"), 29552, extra=("--load-key", key))
    assert loaded["key_loaded_from_file"] and not made["key_loaded_from_file"] and loaded["every_rank_wrote_the_same_bytes"]
    assert open(tmp_path / "p2b.bin", "rb").read() == open(tmp_path / "p2a.bin", "rb").read() == open(tmp_path / "p1.bin", "rb").read()
    assert loaded["sha256"] == one["sha256"]
    # the verifying key gathered from the loaded shares is the keygen run's
    with np.load(str(tmp_path / "p2b.bin") + ".vk.npz", allow_pickle=False) as a, np.load(str(tmp_path / "p1.bin") + ".vk.npz", allow_pickle=False) as b:
        for name in a.files:
            if name.startswith("fixed_"):
                assert np.array_equal(a[name], b[name]), name
    # a one-rank run cannot take rank 0's share of a two-rank key
    os.replace(f"{key}.rank0of2.npz", f"{key}.rank0of1.npz")
    env = dict(os.environ, VDB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sharded_prove.py"), "--circuit", circuit, "--load-key", key], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode != 0 and "proving key does not describe this circuit" in res.stderr


def test_comm_over_rccl_on_the_device():
    """dist.Comm with the nccl (RCCL) backend, as many ranks as the box has GPUs (one here): the collectives' tensors live in HBM
    and the field sum moves the library's own device allocations through torch tensors — the interop the multi-GPU run depends on"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29563", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "comm_nccl_probe.py")], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-1000:], res.stderr[-3000:])
    rep = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert rep == {"backend": "nccl", "world": 1, "sum_disjoint": True, "gather_rows": True, "sum_field_dev": True}
