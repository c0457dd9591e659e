"""cProfile of one untimed whole proof (C4' cosine), host side: where the wall time outside the device goes"""
import cProfile, pstats, sys, time, io
sys.path.insert(0, ".")
from halo2_vectordb_amd import api
from halo2_vectordb_amd.pipeline import KmeansHotPath
from halo2_vectordb_amd.rounds import ProverRounds
I = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hp = KmeansHotPath(n=256, dim=128, K=4, I=I, k=16, P=48, L=15, metric="cosine")
hp.ext_block_cols = 256
hp.setup()
pr = ProverRounds(hp).keygen()
pr.prove(None)
t0 = time.time(); pr.prove(None); print("wall", time.time() - t0, pr.host_ms)
prof = cProfile.Profile()
prof.enable(); pr.prove(None); prof.disable()
s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue())
s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(40); print(s.getvalue())
# per-kernel device time of one proof (HIP events around every launch, read after the proof: nothing is serialised)
api.profile_begin(deferred=True)
pr.prove(None)
api.sync()
rows = api.profile_end()
print("KERNELS", rows if not isinstance(rows, dict) else "")
if isinstance(rows, dict):
    for name, v in sorted(rows.items(), key=lambda kv: -(kv[1]["ms"] if isinstance(kv[1], dict) else kv[1])):
        print(name, v)
