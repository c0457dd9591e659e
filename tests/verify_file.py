#!/usr/bin/env python3
"""Stand-alone check of a proof file written by tools/rounds_bench.py (io.write_snark) against its verifying-key file, on the CPU:
the verifier of tests/test_gpu_rounds.py (_verify: transcript replay from the proof bytes, quotient identity, SHPLONK pairing
equation with oracle/pairing.py).  Test infrastructure (uses the oracle); needs no GPU.
A key file that does not end in .npz is read as halo2's own layout (io.read_verifying_key_raw: what the reference's Verify arm reads,
src/scaffold/mod.rs:298-320) with the SRS scalar of the reference's gen_srs unless one is given.
usage: python tests/verify_file.py proof.snark [proof.snark.vk.npz | name.vk [tau]]"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def _fr_int(O, v):
    return int(O.fr_to_ints(v.reshape(1, 4))[0])


def main(proof_path, vk_path=None, tau=None):
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.io import read_snark, read_verifying_key, read_verifying_key_raw
    from oracle import oracle as O
    from oracle import pairing as PR
    from test_gpu_rounds import _verify, _vk_digest
    proof, instances = read_snark(proof_path)
    if vk_path is not None and not str(vk_path).endswith(".npz"):
        if tau is None:
            from halo2_vectordb_amd.srs import gen_srs_tau
            tau = gen_srs_tau()
        meta, fixed, _selectors = read_verifying_key_raw(vk_path, n_instances=len(instances), tau=int(tau))
        opened = meta.pop("opened")
        tau = meta.pop("tau")
    else:
        meta, fixed = read_verifying_key(vk_path or proof_path + ".vk.npz")
        opened = meta.pop("opened")
        tau = meta.pop("tau")
        digest = meta.pop("vk_digest")
        assert _fr_int(O, _vk_digest(api, fixed)) == digest, "the key's digest is not the digest of its commitments"
    vk = dict(meta=meta, opened=opened, fixed=fixed, tau_h=PR.pt_mul(PR.G2, tau), instances=instances)
    t0 = time.time()
    ok = _verify(O, api, proof, vk)
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 2
    return dict(proof_bytes=len(proof), columns=meta["n_cols"], accepted=bool(ok), tampered_byte_accepted=bool(_verify(O, api, bytes(bad), vk)),
                verify_s=round(time.time() - t0, 1))


if __name__ == "__main__":
    print(json.dumps(main(*sys.argv[1:4])))
