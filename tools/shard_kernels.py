import json, sys
sys.path.insert(0, ".")
from halo2_vectordb_amd import api
from halo2_vectordb_amd.pipeline import KmeansHotPath
api.init(0)
r, w = int(sys.argv[1]), int(sys.argv[2])
hp = KmeansHotPath(col_shard=(r, w)).setup()
hp.step(); hp.step()
api.profile_begin(deferred=True)
for _ in range(3):
    hp.step()
api.sync()
prof = api.profile_end()
rows = sorted(((k, v["ms"] / 3, v["launches"] // 3) for k, v in prof.items()), key=lambda x: -x[1])
print(json.dumps({"rank": r, "world": w, "kernels_ms_per_step": {k: [round(ms, 3), n] for k, ms, n in rows if ms > 0.02}}))
