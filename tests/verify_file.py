#!/usr/bin/env python3
"""Stand-alone check of a proof file written by tools/rounds_bench.py (io.write_snark) against its verifying-key file, on the CPU:
the verifier of tests/test_gpu_rounds.py (_verify: transcript replay from the proof bytes, quotient identity, SHPLONK pairing
equation with oracle/pairing.py).  Test infrastructure (uses the oracle); needs no GPU.
usage: python tests/verify_file.py proof.snark [proof.snark.vk.npz]"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main(proof_path, vk_path=None):
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.io import read_snark
    from oracle import oracle as O
    from oracle import pairing as PR
    from test_gpu_rounds import FIXED, _verify
    proof, instances = read_snark(proof_path)
    with np.load(vk_path or proof_path + ".vk.npz", allow_pickle=False) as doc:
        meta = json.loads(bytes(doc["meta"]).decode())
        fixed = {name: np.ascontiguousarray(doc["fixed_" + name]) for name in FIXED}
    meta["delta"] = int(meta["delta"])
    meta["instance_pos"] = [tuple(p) for p in meta["instance_pos"]]
    opened = {int(rot): names for rot, names in meta.pop("opened").items()}
    tau = int(meta.pop("tau"))
    vk = dict(meta=meta, opened=opened, fixed=fixed, tau_h=PR.pt_mul(PR.G2, tau), instances=instances)
    t0 = time.time()
    ok = _verify(O, api, proof, vk)
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 2
    return dict(proof_bytes=len(proof), columns=meta["n_cols"], accepted=bool(ok), tampered_byte_accepted=bool(_verify(O, api, bytes(bad), vk)),
                verify_s=round(time.time() - t0, 1))


if __name__ == "__main__":
    print(json.dumps(main(*sys.argv[1:3])))
