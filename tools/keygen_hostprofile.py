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
prof = cProfile.Profile()
prof.enable()
t0 = time.time()
hp = KmeansHotPath(n=256, dim=128, K=4, I=I, k=16, P=48, L=15, metric="cosine")
hp.ext_block_cols = 256
hp.setup()
t1 = time.time()
pr = ProverRounds(hp).keygen()
t2 = time.time()
prof.disable()
print("setup_s", round(t1 - t0, 1), "keygen_s", round(t2 - t1, 1))
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue())
