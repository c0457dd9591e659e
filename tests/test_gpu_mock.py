"""The device-side MockProver (vdb_mock_check_dev): the reference's Mock arm (src/scaffold/mod.rs:263-266:
MockProver::run(k, &circuit, instances).assert_satisfied()) on witnesses as they lie in HBM — gate rows, lookup cells,
copies, constants — against the oracle's own row-by-row check (oracle/gadgets.c orc_check_gates) and against deliberately
broken witnesses."""
import numpy as np
import pytest

import examples_common as E

pytestmark = pytest.mark.gpu
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init()
    return a


def test_c1_euclid_mock_stage(api, O):
    """BASELINE configs[0]: euclidean_distance on two 4-dim vectors, k = 13, LOOKUP_BITS = 12, the Mock stage: the product's
    checker on the GPU-generated witness, next to the oracle's check of its own stream"""
    from halo2_vectordb_amd.copymap import lookup_sources
    a, b = [0.123, 0.456, 1.789, 1.123], [1.123, 0.456, 0.789, 0.123]
    qa, qb = api.quantize([a]), api.quantize([b])
    got = api.wit_distance("euclidean", qa, qb, L=12, selectors=True)
    src = lookup_sources(got["flags"], got["lookup"].shape[0])
    rep = api.mock_check(got["stream"], got["flags"], got["lookup"], 12, lookup_src=src, const_stream=got["stream"])
    assert rep.violations() == 0, rep.as_dict()
    c = O.Ctx(store=True, keygen=True, plan_k=13)
    r = c.distance("euclidean", qa[0], qb[0], L=12)
    assert c.check_gates(12) == 0 and np.array_equal(got["result"][0], r)
    n_gates = int((got["flags"] & 1).sum())
    assert n_gates > 4000
    # broken witnesses: one cell of a gate, one lookup cell pushed out of the table, one lookup cell that no longer is the cell it copies
    gate_rows = np.flatnonzero(got["flags"] & 1)
    for row in (gate_rows[0], gate_rows[len(gate_rows) // 2], gate_rows[-1]):
        bad = got["stream"].copy()
        bad[row + 3] = O.fr_add(bad[row + 3].reshape(1, 4), O.fr_from_ints([1]))[0]
        rep = api.mock_check(bad, got["flags"], got["lookup"], 12)
        assert rep.gate_rows_violated >= 1 and rep.first_gate_row <= row
    badl = got["lookup"].copy()
    badl[7] = O.fr_from_ints([1 << 12])[0]
    rep = api.mock_check(got["stream"], got["flags"], badl, 12, lookup_src=src)
    assert rep.lookup_cells_out_of_table == 1 and rep.first_lookup_cell == 7 and rep.lookup_copies_unequal == 1 and rep.gate_rows_violated == 0
    badl = got["lookup"].copy()
    badl[9] = O.fr_from_ints([(O.fr_to_ints(badl[9].reshape(1, 4))[0] + 1) % 4096])[0]     # still in the table, no longer the copied cell
    rep = api.mock_check(got["stream"], got["flags"], badl, 12, lookup_src=src)
    assert rep.lookup_cells_out_of_table == 0 and rep.lookup_copies_unequal == 1 and rep.first_lookup_copy == 9
    # a constant cell altered: gates may still hold (a constant is one operand among three), the constant check does not
    cst = np.flatnonzero(got["flags"] & 2)
    bad = got["stream"].copy()
    bad[cst[5]] = O.fr_from_ints([12345])[0]
    rep = api.mock_check(bad, got["flags"], got["lookup"], 12, const_stream=got["stream"])
    assert rep.constants_changed == 1 and rep.first_constant == cst[5]
    # a copy map: cell 40 declared a copy of cell 3
    copy_of = np.arange(got["stream"].shape[0], dtype=np.int64)
    copy_of[40] = 3
    rep = api.mock_check(got["stream"], got["flags"], got["lookup"], 12, copy_of=copy_of)
    assert rep.copies_unequal == (0 if np.array_equal(got["stream"][40], got["stream"][3]) else 1)


def test_query_example_ties_violate_select_by_indicator(api, O):
    """examples/query.rs on data/query.in: the database's duplicate rows tie, and select_by_indicator's witness rule breaks its
    own accumulation gate on every extra tie (tests/test_examples_cpu.py::test_query_in): the GPU checker finds the same
    12 rows in the GPU's witness as the oracle's checker does in the oracle's"""
    r = E.oracle_query(O)
    d = E.load("query")
    qq, qdb = api.quantize(np.array(d["query"])), api.quantize(np.array(d["database"]))
    nv = api.wit_nearest("cosine", qq, qdb, L=12, selectors=True)
    rep = api.mock_check(nv["stream"], nv["flags"], nv["lookup"], 12)
    assert rep.gate_rows_violated == 12 == r["ctx"].check_gates(12) and rep.lookup_cells_out_of_table == 0
    # without the duplicates the same circuit is satisfied
    uniq = qdb[:4]
    nv = api.wit_nearest("cosine", qq, uniq, L=12, selectors=True)
    assert api.mock_check(nv["stream"], nv["flags"], nv["lookup"], 12).violations() == 0


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_hot_path_mock_stage_on_resident_buffers(api, O, metric):
    """the checker on the hot path's resident streams (no download): a k-means circuit at 2^12 rows, with the lookup cells
    tied to the advice cells they copy and the constants compared with the keygen-time stream"""
    from halo2_vectordb_amd.copymap import lookup_sources
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    hp = KmeansHotPath(n=24, dim=8, K=3, I=2, k=12, P=48, L=11, seed=5, metric=metric).setup()
    hp.step()
    d_flags = hp.keygen_flags()
    flags = d_flags.download((hp.n_cells,), dtype=np.uint8)
    d_flags.free()
    src = lookup_sources(flags, hp.n_lookup)
    keygen_stream = hp.d_stream.download((hp.n_cells, 4))
    # prove another database of the same shape: the constants must not move, every gate must hold
    rng = np.random.default_rng(6)
    hp.set_vectors(rng.integers(0, 219, size=(24, 8)).astype(np.float64) + rng.random((24, 8)))
    rep = hp.mock_check(lookup_src=src, const_stream=keygen_stream)
    assert rep.violations() == 0, rep.as_dict()
    # and the same stream broken in HBM
    one = O.fr_from_ints([1])[0]
    row = int(np.flatnonzero(flags & 1)[1000])
    cell = hp.d_stream.download((4,), offset=(row + 3) * 32)
    hp.d_stream.upload(O.fr_add(cell.reshape(1, 4), one.reshape(1, 4))[0], offset=(row + 3) * 32)
    d_flags = api.DeviceBuffer(hp.n_cells)
    d_flags.upload(flags)
    rep = api.mock_check_dev(hp.d_stream.ptr, hp.n_cells, d_flags.ptr, hp.d_lookup.ptr, hp.n_lookup, hp.L)
    d_flags.free()
    assert rep.gate_rows_violated >= 1 and rep.first_gate_row <= row
    hp.free()
