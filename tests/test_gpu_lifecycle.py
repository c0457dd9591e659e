"""Setup -> keygen -> proof -> free, over and over: what a long-lived prover process does.  Device memory in use and the host's peak
resident set must not grow with the number of proofs (the library's work space and tables are grow-only but bounded by the largest
job seen; everything a hot path or a key owns goes back with free())."""
import resource

import pytest

pytestmark = pytest.mark.gpu


def test_repeated_keygen_and_proofs_leave_no_memory_behind():
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.pipeline import DistancesHotPath, KmeansHotPath, MerkleHotPath, QueryHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    api.init(0)

    def used_mib():
        free, total = api.mem_info()
        return (total - free) / 2 ** 20
    seen = []
    for it in range(6):
        for make in (lambda: KmeansHotPath(n=8, dim=4, K=2, I=1, k=12, L=11, metric="cosine"), lambda: MerkleHotPath(n=6, dim=5, k=11),
                     lambda: QueryHotPath(n=5, dim=4, k=12, L=11), lambda: DistancesHotPath(dim=4, k=13, L=12)):
            hp = make().setup()
            pr = ProverRounds(hp).keygen()
            try:
                assert len(pr.prove(None)["proof"]) > 0
            finally:
                pr.free()
                hp.free()
        api.sync()
        seen.append((used_mib(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024))
    # the first rounds size the grow-only work space; after that nothing may move (a MiB of slack for the allocator's own bookkeeping)
    assert abs(seen[-1][0] - seen[2][0]) <= 1.0, seen
    assert seen[-1][1] - seen[2][1] <= 8.0, seen
