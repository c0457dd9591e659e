#!/usr/bin/env python3
"""cProfile of setup + keygen of the cosine k = 16 circuit (C4'): where the ~30 s go (host-side map construction against device work)"""
import cProfile
import io
import pstats
import sys
import time

sys.path.insert(0, ".")
from halo2_vectordb_amd import api
from halo2_vectordb_amd.pipeline import KmeansHotPath
from halo2_vectordb_amd.rounds import ProverRounds

I = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# every device allocation and every commitment with its size and wall time (where keygen + setup go)
ALLOCS, COMMITS = [], []
_init = api.DeviceBuffer.__init__


def _timed_init(self, nbytes):
    t0 = time.perf_counter()
    _init(self, nbytes)
    ALLOCS.append((int(nbytes), time.perf_counter() - t0))


api.DeviceBuffer.__init__ = _timed_init
_commit = ProverRounds._commit


def _timed_commit(self, buf, n_cols, basis, dense=True):
    api.sync()
    t0 = time.perf_counter()
    out = _commit(self, buf, n_cols, basis, dense)
    COMMITS.append((int(n_cols), basis, time.perf_counter() - t0))
    return out


ProverRounds._commit = _timed_commit
prof = cProfile.Profile()
prof.enable()
t0 = time.time()
hp = KmeansHotPath(n=256, dim=128, K=4, I=I, k=16, P=48, L=15, metric="cosine")
hp.ext_block_cols = 256
hp.setup()
api.sync()
t1 = time.time()
pr = ProverRounds(hp).keygen()
api.sync()
t2 = time.time()
prof.disable()
print("setup_s", round(t1 - t0, 1), "keygen_s", round(t2 - t1, 1))
big = sorted(ALLOCS, reverse=True)[:24]
print("device allocations:", len(ALLOCS), "total GB", round(sum(a for a, _ in ALLOCS) / 2**30, 1), "total s", round(sum(t for _, t in ALLOCS), 2))
print("largest (GB, s):", [(round(a / 2**30, 1), round(t, 2)) for a, t in big])
print("commitments (columns, basis, s):", [(n, b, round(t, 2)) for n, b, t in COMMITS])
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue())
