"""The prover rounds after the advice commitments (halo2_vectordb_amd/rounds.py), end to end on a small k-means circuit:
everything a verifier would check, checked — the quotient identity at the evaluation point recombined from the
evaluations, and every opening against its commitments in the exponent (the test knows tau) — once with the challenges
handed in, once from nothing but the proof bytes and the fixed commitments, replaying the Fiat–Shamir transcript.
Parity unpinned: the reference holds no vectors for the prover rounds; these are the PLONK / KZG
identities themselves."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TAU = 0x1234567890ABCDEF1234567
FIXED = ("sel", "sigma", "cst", "table")


def _vk_digest(api, fixed):
    """the verifying key enters the transcript as one scalar, as halo2's vk.transcript_repr does: here the squeeze of a sponge of
    its own over every fixed commitment (a verifier computes it once per key)"""
    tr0 = api.Transcript()
    for name in FIXED:
        for pt in fixed[name]:
            tr0.common_point(pt)
    d = tr0.squeeze()
    tr0.free()
    return d
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


@pytest.fixture(scope="module")
def circuit():
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    api.init(0)
    # the cosine variant: the satisfiable one (examples/kmeans.rs:48-49); Euclidean k-means breaks qlog2's asserted constants
    # at iteration 0, where every initial centroid is at distance 0 of its own vector (tests/test_gpu_copymap.py shows that)
    hp = KmeansHotPath(n=8, dim=4, K=2, I=1, k=12, L=11, metric="cosine", tau=TAU).setup()
    pr = ProverRounds(hp).keygen()
    assert pr.keygen_report.violations() == 0, pr.keygen_report.as_dict()
    yield hp, pr
    pr.free()
    hp.free()


@pytest.fixture(scope="module")
def proved(circuit, O):
    hp, pr = circuit
    rng = np.random.default_rng(99)
    ch = {name: O.random_fr(rng, 1)[0] for name in ("beta", "gamma", "y", "x", "v")}
    timings = {}
    out = pr.prove(ch, seed=5, timings=timings, multiopen="gwc")
    return ch, out, timings


def instance_poly_at(O, instances, x, k):
    """the instance column at x from the public values alone: sum_i v_i L_i(x), L_i(x) = w^i (x^n - 1) / (n (x - w^i)) — the
    verifier's side of an uncommitted instance column (halo2: QUERY_INSTANCE = false)"""
    R, n = O.R_MOD, 1 << k
    w = O.fr_to_ints(O.root_of_unity(k).reshape(1, 4))[0]
    acc = 0
    for i, v in enumerate(instances):
        wi = pow(w, i, R)
        acc = (acc + v * wi * pow((x - wi) % R, -1, R)) % R
    return acc * (pow(x, n, R) - 1) * pow(n, -1, R) % R


def folded_h(O, meta, ch, evals, instances, h_commits):
    """What a verifier derives instead of reading it (halo2's vanishing argument): the value of h folded at x — the quotient
    identity's numerator over x^n - 1 — and the folded commitment sum_i [x^(n i)] H_i.  None when the statement is malformed."""
    R = O.R_MOD
    num = quotient_numerator(O, meta, ch, evals, instances)
    if num is None:
        return None
    x = O.fr_to_ints(np.asarray(ch["x"]).reshape(1, 4))[0]
    xn = pow(x, meta["rows"], R)
    value = num * pow((xn - 1) % R, -1, R) % R
    commit = O.msm_naive(O.fr_from_ints([pow(xn, i, R) for i in range(len(h_commits))]), np.asarray(h_commits))
    return value, commit


def check_quotient_identity(O, meta, ch, evals, instances=()):
    """gates + permutation + lookup expressions from the evaluations == hf(x) (x^n - 1), hf(x) = the prover's own evaluation of h
    folded at x (returned by prove(); a verifier does not receive it: folded_h)"""
    R = O.R_MOD
    num = quotient_numerator(O, meta, ch, evals, instances)
    if num is None or not evals.get(("hf", 0)):
        return False
    x = O.fr_to_ints(np.asarray(ch["x"]).reshape(1, 4))[0]
    return num == evals[("hf", 0)][0] * (pow(x, meta["rows"], R) - 1) % R and num != 0


def quotient_numerator(O, meta, ch, evals, instances=()):
    """gates + permutation + lookup expressions recombined from the evaluations (None: wrong number of public values, or evaluations
    missing where this key's shape asks for them — a key that does not describe the proof's circuit)"""
    try:
        return _quotient_numerator(O, meta, ch, evals, instances)
    except IndexError:
        return None


def _quotient_numerator(O, meta, ch, evals, instances=()):
    R = O.R_MOD
    to_int = lambda a: O.fr_to_ints(np.asarray(a).reshape(1, 4))[0]
    b, g, yv, x = (to_int(ch[n]) for n in ("beta", "gamma", "y", "x"))
    delta, n, n_adv, chunk, n_blind = meta["delta"], meta["rows"], meta["n_adv"], meta["chunk_len"], meta["n_blind"]
    ev = lambda name, rot=0: evals.get((name, rot), [])
    acc = 0
    a0, a1, a2, a3, q = ev("adv"), ev("advg", 1), ev("advg", 2), ev("advg", 3), ev("sel")
    for c in range(n_adv):
        acc = (acc * yv + q[c] * (a0[c] + a1[c] * a2[c] - a3[c])) % R
    if len(instances) != meta["n_instances"]:
        return None
    # l_0, l_last, l_active = 1 - l_last - l_blind at x: not in the proof, a verifier computes them from the domain
    w_n = O.fr_to_ints(np.asarray(O.root_of_unity(meta["k"])).reshape(1, 4))[0]
    zn = (pow(x, n, R) - 1) * pow(n, -1, R) % R
    l_at = lambda i: pow(w_n, i, R) * zn % R * pow((x - pow(w_n, i, R)) % R, -1, R) % R
    usable = n - n_blind
    l0, ll = l_at(0), l_at(usable)
    la = (1 - ll - sum(l_at(i) for i in range(usable + 1, n))) % R
    sg, z0, z1, zb = ev("sigma"), ev("zp"), ev("zp", 1), ev("zp", -n_blind)
    # the permutation's columns: advice, lookup, the constants' fixed column, the instance column (from the public values)
    pcols = list(a0) + list(ev("cst")) + [instance_poly_at(O, instances, x, meta["k"])]
    n_cols, n_sets = len(pcols), len(z0)
    if n_cols != meta["n_cols"] + 2 or len(sg) != n_cols:
        return None
    acc = (acc * yv + l0 * (1 - z0[0])) % R
    acc = (acc * yv + ll * (z0[-1] * z0[-1] - z0[-1])) % R
    for i in range(1, n_sets):
        acc = (acc * yv + l0 * (z0[i] - zb[i - 1])) % R
    cur = b * x % R
    for i in range(n_sets):
        left, right = z1[i], z0[i]
        for c in range(i * chunk, min((i + 1) * chunk, n_cols)):
            left = left * (pcols[c] + b * sg[c] + g) % R
            right = right * (pcols[c] + cur + g) % R
            cur = cur * delta % R
        acc = (acc * yv + la * (left - right)) % R
    A, S, PA, PS, PAm, Z, Z1 = a0[n_adv:], ev("table")[0], ev("pa"), ev("ps"), ev("pa", -1), ev("zl"), ev("zl", 1)
    for c in range(len(A)):
        acc = (acc * yv + l0 * (1 - Z[c])) % R
        acc = (acc * yv + ll * (Z[c] * Z[c] - Z[c])) % R
        acc = (acc * yv + la * (Z1[c] * (PA[c] + b) * (PS[c] + g) - Z[c] * (A[c] + b) * (S + g))) % R
        acc = (acc * yv + l0 * (PA[c] - PS[c])) % R
        acc = (acc * yv + la * (PA[c] - PS[c]) * (PA[c] - PAm[c])) % R
    return acc


def check_openings(O, v_int, commitments, evals, openings):
    """sum_i v^(m-1-i) C_i - [eval] G == [tau - point] W for each rotation point"""
    R = O.R_MOD
    G = O.g1_generator().reshape(1, 8)
    for op in openings:
        commits = np.concatenate([commitments[name] for name in op["polys"]])
        evs = [e for name in op["polys"] for e in evals[(name, op["rotation"])]]
        m = len(evs)
        if commits.shape[0] != m:
            return False
        comb = 0
        for e in evs:
            comb = (comb * v_int + e) % R
        scalars = [pow(v_int, m - 1 - i, R) for i in range(m)] + [(-comb) % R]
        lhs = O.msm_naive(O.fr_from_ints(scalars), np.concatenate([commits, G]))
        rhs = O.msm_naive(O.fr_from_ints([(TAU - op["point"]) % R]), op["W"].reshape(1, 8))
        if not (np.array_equal(lhs, rhs) and lhs.any()):
            return False
    return True


def _meta(pr):
    from halo2_vectordb_amd.rounds import N_BLIND
    import halo2_vectordb_amd.rounds as rounds
    return dict(rows=pr.rows, k=pr.k, n_adv=pr.n_adv, n_lk=pr.n_lk, n_cols=pr.n_cols, n_sets=pr.n_sets, chunk_len=pr.chunk_len, n_blind=N_BLIND,
                delta=rounds._fr_to_int(pr.delta), n_instances=len(pr.instance_cells))


def test_round_outputs_have_the_expected_shape(circuit, proved):
    hp, pr = circuit
    ch, out, timings = proved
    assert pr.n_adv >= 2 and pr.n_lk >= 1
    c = out["commitments"]
    CHUNK_LEN = pr.chunk_len
    assert CHUNK_LEN == 2 and pr.degree == 4   # cs.degree() = 4 for the vertical gate + single-column lookups (d - 2 = 2)
    assert c["adv"].shape == (pr.n_cols, 8) and c["zp"].shape == (pr.n_sets, 8) and c["h"].shape == (CHUNK_LEN + 1, 8) and c["rand"].shape == (1, 8) and c["hf"].shape == (1, 8) and c["pa"].shape == (pr.n_lk, 8)
    assert len(out["openings"]) == 6 and out["proof"] is None
    for name in ("witness", "commit_msm", "ntt", "lookup_permute", "products", "quotient", "evaluations", "openings"):
        assert timings[name] > 0


def test_quotient_identity_from_the_returned_evaluations(circuit, proved, O):
    from halo2_vectordb_amd.rounds import quotient_identity_holds
    ch, out, _ = proved
    inst = out["instances"]
    assert len(inst) == 2 * 4                                                     # the K x dim centroid words (examples/kmeans.rs:51-56)
    assert check_quotient_identity(O, _meta(circuit[1]), ch, out["evals"], inst)
    assert quotient_identity_holds(circuit[1], out["challenges"], out["evals"], inst)      # the library-side helper agrees
    # another centroid word than the one the witness holds: the instance column's value at x changes, the identity breaks
    assert not check_quotient_identity(O, _meta(circuit[1]), ch, out["evals"], inst[:3] + [(inst[3] + 1) % O.R_MOD] + inst[4:])
    assert not check_quotient_identity(O, _meta(circuit[1]), ch, out["evals"], inst[:-1])
    wrong = dict(out["evals"])
    wrong[("zl", 1)] = [(e + 1) % O.R_MOD for e in wrong[("zl", 1)]]
    assert not check_quotient_identity(O, _meta(circuit[1]), ch, wrong, inst)
    # the constants' fixed column is one of the permutation's columns: another evaluation of it breaks the identity
    assert len(out["evals"][("cst", 0)]) == 1 and len(out["evals"][("sigma", 0)]) == circuit[1].n_cols + 2
    wrong = dict(out["evals"])
    wrong[("cst", 0)] = [(e + 1) % O.R_MOD for e in wrong[("cst", 0)]]
    assert not check_quotient_identity(O, _meta(circuit[1]), ch, wrong, inst)


def test_every_opening_verifies_in_the_exponent(circuit, proved, O):
    ch, out, _ = proved
    v = O.fr_to_ints(ch["v"].reshape(1, 4))[0]
    for op in out["openings"]:     # the combined evaluation the division returned is the combination of the individual ones
        comb = 0
        for name in op["polys"]:
            for e in out["evals"][(name, op["rotation"])]:
                comb = (comb * v + e) % O.R_MOD
        assert comb == O.fr_to_ints(op["eval"].reshape(1, 4))[0]
    assert check_openings(O, v, out["commitments"], out["evals"], out["openings"])
    bad = dict(out["evals"])
    bad[("advg", 2)] = [(e + 1) % O.R_MOD for e in bad[("advg", 2)]]
    assert not check_openings(O, v, out["commitments"], bad, out["openings"])


def test_advice_commitments_are_the_hot_path_commitments(circuit, proved, O):
    hp, pr = circuit
    _, out, _ = proved
    hp.relayout()
    cols = hp.download_columns([0, pr.n_adv - 1, pr.n_adv])
    want = O.msm_batch(cols, hp.g_lagrange)
    assert np.array_equal(out["commitments"]["adv"][[0, pr.n_adv - 1, pr.n_adv]], want)


def _decompress(O, enc_bytes):
    enc = int.from_bytes(enc_bytes, "little")
    if enc == 0:
        return np.zeros(8, dtype=np.uint64)
    x, odd = enc & ((1 << 254) - 1), (enc >> 254) & 1
    y = pow((x * x * x + 3) % Q_MOD, (Q_MOD + 1) // 4, Q_MOD)
    assert (y * y - x * x * x - 3) % Q_MOD == 0
    if (y & 1) != odd:
        y = Q_MOD - y
    return O.fq_from_ints([x, y]).reshape(8)


def test_proof_bytes_verify_from_the_fixed_commitments_alone(circuit, O):
    """prove with the Fiat–Shamir transcript; then a verifier that sees only the proof bytes, the fixed commitments and the
    circuit's shape: parse, replay the transcript for the challenges, check the quotient identity and every opening"""
    from halo2_vectordb_amd import api
    hp, pr = circuit
    assert pr.n_cols > 2 * 30
    pr.block_cols = 30            # the selector / sigma cosets in several blocks (sets 0-9, 10-19, ...)
    out = pr.prove(None, seed=11, multiopen="gwc")
    pr.block_cols = 510
    proof, meta, opened = out["proof"], _meta(pr), out["opened"]
    fixed = {name: pr.fixed[name].commits for name in FIXED}
    counts = {"adv": meta["n_cols"], "advg": meta["n_adv"], "sel": meta["n_adv"], "sigma": meta["n_cols"] + 2, "cst": 1, "table": 1, "pa": meta["n_lk"], "ps": meta["n_lk"],
              "zp": meta["n_sets"], "zl": meta["n_lk"], "rand": 1, "hf": 0}          # (hf: opened, but its evaluation is not sent)
    n_evals = sum(counts[name] for names in opened.values() for name in names)
    n_points = meta["n_cols"] + 2 * meta["n_lk"] + meta["n_sets"] + meta["n_lk"] + 1 + (meta["chunk_len"] + 1) + len(opened)
    assert len(proof) == 32 * (n_points + n_evals)

    # ---- the verifier
    pos = 0
    def take_points(m):
        nonlocal pos
        pts = np.stack([_decompress(O, proof[pos + 32 * i: pos + 32 * i + 32]) for i in range(m)])
        pos += 32 * m
        return pts
    tr = api.Transcript()
    tr.common_scalar(_vk_digest(api, fixed))
    for value in out["instances"]:                  # the statement: the public centroid words
        tr.common_scalar(O.fr_from_ints([value])[0])
    commitments = dict(fixed)
    def absorb(pts):
        for pt in pts:
            tr.common_point(pt)
        return pts
    commitments["adv"] = absorb(take_points(counts["adv"]))
    commitments["advg"] = commitments["adv"][: meta["n_adv"]]          # the gate columns, opened at rows 1..3 as a group of their own
    ch = {"theta": tr.squeeze()}
    pairs = absorb(take_points(2 * meta["n_lk"]))
    commitments["pa"], commitments["ps"] = pairs[0::2], pairs[1::2]
    ch["beta"], ch["gamma"] = tr.squeeze(), tr.squeeze()
    commitments["zp"] = absorb(take_points(counts["zp"]))
    commitments["zl"] = absorb(take_points(counts["zl"]))
    commitments["rand"] = absorb(take_points(1))                       # the vanishing argument's random polynomial, before y
    ch["y"] = tr.squeeze()
    commitments["h"] = absorb(take_points(meta["chunk_len"] + 1))      # degree - 1 pieces, chunk_len = degree - 2
    ch["x"] = tr.squeeze()
    evals = {}
    for rot, names in opened.items():
        for name in names:
            vals = []
            for _ in range(counts[name]):
                e = int.from_bytes(proof[pos: pos + 32], "little")
                pos += 32
                assert e < O.R_MOD
                tr.common_scalar(O.fr_from_ints([e])[0])
                vals.append(e)
            evals[(name, rot)] = vals
    ch["v"] = tr.squeeze()
    Ws = take_points(len(opened))
    assert pos == len(proof)
    tr.free()
    x = O.fr_to_ints(ch["x"].reshape(1, 4))[0]
    w = O.fr_to_ints(O.root_of_unity(meta["k"]).reshape(1, 4))[0]
    openings = [dict(rotation=rot, polys=names, point=x * pow(w, rot % meta["rows"], O.R_MOD) % O.R_MOD, W=Ws[i]) for i, (rot, names) in enumerate(opened.items())]
    # the challenges the prover used are the ones the proof bytes determine
    for name in ("beta", "gamma", "y", "x", "v"):
        assert np.array_equal(ch[name], out["challenges"][name])
    # h folded at x: value from the quotient identity, commitment from the pieces' — the opening at x then checks the identity itself
    hf_value, hf_commit = folded_h(O, meta, ch, evals, out["instances"], commitments["h"])
    evals[("hf", 0)], commitments["hf"] = [hf_value], hf_commit.reshape(1, 8)
    assert hf_value == out["evals"][("hf", 0)][0] and np.array_equal(commitments["hf"], out["commitments"]["hf"])
    assert check_openings(O, O.fr_to_ints(ch["v"].reshape(1, 4))[0], commitments, evals, openings)
    evals[("hf", 0)] = [(hf_value + 1) % O.R_MOD]
    assert not check_openings(O, O.fr_to_ints(ch["v"].reshape(1, 4))[0], commitments, evals, openings)


def test_rounds_on_a_merkle_circuit_without_lookups(O):
    """the same rounds on merkle_commitment (Poseidon trace, no lookup columns, every column dense): proof bytes from the
    transcript, quotient identity and openings checked from the returned values"""
    from halo2_vectordb_amd.pipeline import MerkleHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    hp = MerkleHotPath(n=6, dim=5, k=11, tau=TAU).setup()      # 6 leaves (padded to 8), 3 + 1 permutations per leaf
    pr = ProverRounds(hp).keygen()
    try:
        assert pr.n_lk == 0 and pr.n_adv >= 3
        out = pr.prove(None, seed=3, multiopen="gwc")
        meta = _meta(pr)
        assert out["commitments"]["pa"].shape == (0, 8) and len(out["proof"]) > 0
        assert check_quotient_identity(O, meta, out["challenges"], out["evals"], out["instances"])
        v = O.fr_to_ints(out["challenges"]["v"].reshape(1, 4))[0]
        assert check_openings(O, v, out["commitments"], out["evals"], out["openings"])
        # and the SHPLONK proof bytes through the stand-alone verifier (defined below)
        from halo2_vectordb_amd import api
        from oracle import pairing as PR
        out2 = pr.prove(None, seed=4)
        vk = dict(meta=_meta(pr), opened=out2["opened"], fixed={name: pr.fixed[name].commits for name in FIXED},
                  tau_h=PR.pt_mul(PR.G2, TAU), instances=out2["instances"])
        assert _verify(O, api, out2["proof"], vk)
        assert not _verify(O, api, out2["proof"], {**vk, "instances": [(out2["instances"][0] + 1) % O.R_MOD]})    # another root: rejected
        # the key files in upstream's layout at degree 3 (an extended domain of 2n points): the key read back proves the same bytes,
        # the .vk file alone tells the verifier the circuit has no lookup columns
        import tempfile
        import verify_file
        from halo2_vectordb_amd.io import write_snark
        with tempfile.TemporaryDirectory() as d:
            pr.save_proving_key_raw(os.path.join(d, "merkle.pk"))
            pr.save_verifying_key_raw(os.path.join(d, "merkle.vk"))
            assert os.path.getsize(os.path.join(d, "merkle.pk")) == os.path.getsize(os.path.join(d, "merkle.vk")) + 3 * (4 + 64 * pr.rows) \
                + (pr.n_adv + 2 + pr.n_perm) * (2 * (4 + 32 * pr.rows) + 4 + 64 * pr.rows) + 24
            pr2 = ProverRounds(hp).load_proving_key_raw(os.path.join(d, "merkle.pk"))
            try:
                assert pr2.prove(None, seed=4)["proof"] == out2["proof"]
            finally:
                pr2.free()
            write_snark(os.path.join(d, "merkle.snark"), out2["proof"], out2["instances"])
            rep = verify_file.main(os.path.join(d, "merkle.snark"), os.path.join(d, "merkle.vk"), TAU)
            assert rep["accepted"] and not rep["tampered_byte_accepted"]
    finally:
        pr.free()
        hp.free()


def _interpolate(pts, vals, R):
    coeffs = [0] * len(pts)
    for i, (xi, yi) in enumerate(zip(pts, vals)):
        basis, denom = [1], 1
        for j, xj in enumerate(pts):
            if j != i:
                basis = [(a - xj * b) % R for a, b in zip([0] + basis, basis + [0])]
                denom = denom * (xi - xj) % R
        scale = yi * pow(denom, -1, R) % R
        coeffs = [(c + scale * b) % R for c, b in zip(coeffs, basis)]
    return coeffs


def test_shplonk_multiopen_verifies_in_the_exponent(circuit, O):
    """the multi-open the reference's gen_snark_shplonk runs: two commitments for all rotation sets.  The verifier's side,
    from the proof's commitments and evaluations and the challenges:
        sum_S v^(m-1-s) Z_{T - S}(u) (C_S - [r_S(u)] G) - Z_T(u) W1 == [tau - u] W2,   C_S = the set's commitments combined with yo"""
    from halo2_vectordb_amd.rounds import quotient_identity_holds
    hp, pr = circuit
    out = pr.prove(None, seed=21)            # multiopen="shplonk" is the default
    op, R = out["openings"], O.R_MOD
    assert op["kind"] == "shplonk" and all(r == 0 for r in op["remainders"]) and len(op["remainders"]) >= 2
    assert quotient_identity_holds(pr, out["challenges"], out["evals"], out["instances"])
    assert len(out["proof"]) % 32 == 0
    to_int = lambda a: O.fr_to_ints(np.asarray(a).reshape(1, 4))[0]
    yo, v, u = (to_int(out["challenges"][n]) for n in ("yo", "v", "u"))
    pts = out["points"]
    sets = op["sets"]
    m = len(sets)
    assert sorted(len(rots) for rots, _ in sets) == [1, 2, 2, 3, 3]     # fixed, h & every advice column at x; pa; zl; zp; the gate columns at rows 1..3
    all_rots = sorted({rot for rots, _ in sets for rot in rots})
    def vanish(rots, x):
        acc = 1
        for rot in rots:
            acc = acc * (x - pts[rot]) % R
        return acc
    scalars, bases, g_scalar = [], [], 0
    for s_i, (rots, names) in enumerate(sets):
        commits = np.concatenate([out["commitments"][name] for name in names])
        k_cols = commits.shape[0]
        vals = []
        for rot in rots:
            acc = 0
            for name in names:
                for e in out["evals"][(name, rot)]:
                    acc = (acc * yo + e) % R
            vals.append(acc)
        r = _interpolate([pts[rot] for rot in rots], vals, R)
        r_u = 0
        for c in reversed(r):
            r_u = (r_u * u + c) % R
        coef = pow(v, m - 1 - s_i, R) * vanish([rot for rot in all_rots if rot not in rots], u) % R
        scalars += [coef * pow(yo, k_cols - 1 - i, R) % R for i in range(k_cols)]
        bases.append(commits)
        g_scalar = (g_scalar - coef * r_u) % R
    scalars += [g_scalar, (-vanish(all_rots, u)) % R]
    bases += [O.g1_generator().reshape(1, 8), op["W1"].reshape(1, 8)]
    lhs = O.msm_naive(O.fr_from_ints(scalars), np.concatenate(bases))
    rhs = O.msm_naive(O.fr_from_ints([(TAU - u) % R]), op["W2"].reshape(1, 8))
    assert np.array_equal(lhs, rhs) and lhs.any()
    # the same statement the way a verifier checks it, without the toxic scalar: e(C_L + [u] W2, H) e(-W2, [tau] H) == 1 with
    # [tau] H from the G2 side of the SRS (oracle/pairing.py)
    from oracle import pairing as PR
    to_pt = lambda a: None if not np.asarray(a).any() else tuple(O.fq_to_ints(np.asarray(a).reshape(2, 4)))
    left = O.msm_naive(O.fr_from_ints(scalars + [u]), np.concatenate(bases + [op["W2"].reshape(1, 8)]))
    tau_h = PR.pt_mul(PR.G2, TAU)
    assert PR.pairing_product_is_one([(to_pt(left), PR.G2), (PR.pt_neg(to_pt(op["W2"])), tau_h)])
    assert not PR.pairing_product_is_one([(to_pt(left), PR.G2), (PR.pt_neg(to_pt(op["W1"])), tau_h)])
    # a wrong evaluation in the proof breaks the equation
    bad = (g_scalar + 1) % R
    lhs_bad = O.msm_naive(O.fr_from_ints(scalars[:-2] + [bad, scalars[-1]]), np.concatenate(bases))
    assert not np.array_equal(lhs_bad, rhs)


def test_proving_key_round_trip_gives_the_same_proof(circuit, tmp_path):
    """Keygen arm -> file -> Prove arm: a second ProverRounds that never ran keygen, loaded from the saved proving key,
    produces byte-identical proofs; a key for another circuit is refused"""
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    hp, pr = circuit
    path = tmp_path / "kmeans.pk.npz"
    pr.save_proving_key(path)
    want = pr.prove(None, seed=77)["proof"]
    pr2 = ProverRounds(hp).load_proving_key(path)
    try:
        assert all(np.array_equal(pr2.fixed[name].commits, pr.fixed[name].commits) for name in pr.fixed)
        got = pr2.prove(None, seed=77)["proof"]
        assert got == want and len(got) > 0
    finally:
        pr2.free()
    other = KmeansHotPath(n=8, dim=4, K=2, I=1, k=12, L=10, metric="cosine", tau=TAU).setup()      # another lookup width: another circuit
    try:
        with pytest.raises(ValueError):
            ProverRounds(other).load_proving_key(path)
    finally:
        other.free()


def test_key_files_in_upstreams_layout(circuit, O, tmp_path):
    """Keygen arm -> data/{name}.vk + data/{name}.pk in halo2's SerdeFormat::RawBytes layout (src/scaffold/mod.rs:273-281; io.py restates
    the layout, [UPSTREAM-RECALL]) -> Prove arm (custom_read_pk, :325-331) -> Verify arm (custom_read_vk, :334-343): the layout field by
    field, the same proof bytes from the loaded key, the proof accepted against the .vk file, another circuit's key refused."""
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.io import read_verifying_key_raw, write_snark
    from halo2_vectordb_amd.pipeline import KmeansHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    import verify_file
    hp, pr = circuit
    vk_path, pk_path = str(tmp_path / "kmeans.vk"), str(tmp_path / "kmeans.pk")
    pr.save_verifying_key_raw(vk_path)
    pr.save_proving_key_raw(pk_path)
    raw = open(vk_path, "rb").read()
    n, ne, n_fixed = pr.rows, pr.rows * 4, pr.n_adv + 2        # halo2's extended domain: 4n points at degree 4
    pts = lambda a: np.ascontiguousarray(a, dtype="<u8").tobytes()
    assert raw[:4] == pr.k.to_bytes(4, "big") and raw[4:8] == n_fixed.to_bytes(4, "big")
    body = pts(pr.fixed["table"].commits) + pts(pr.fixed["cst"].commits) + pts(pr.fixed["sel"].commits) + pts(pr.fixed["sigma"].commits)
    assert raw[8: 8 + len(body)] == body and len(raw) == 8 + len(body) + pr.n_adv * n // 8
    meta, fixed, selectors = read_verifying_key_raw(vk_path, n_instances=len(pr.instance_cells))
    assert {key: meta[key] for key in _meta(pr)} == _meta(pr)
    assert all(np.array_equal(fixed[name], pr.fixed[name].commits) for name in FIXED)
    # the selectors' bits: one per gate of the keygen witness, where the selector columns of the proving key are one
    flags = hp.keygen_flags()
    assert selectors.shape == (pr.n_adv, n) and int(selectors.sum()) == int((flags.download((hp.n_cells,), dtype=np.uint8) & 1).sum())
    flags.free()
    with open(pk_path, "rb") as f:
        assert f.read(len(raw)) == raw                                     # the proving key starts with the verifying key
        poly = lambda m: (int.from_bytes(f.read(4), "big"), np.frombuffer(f.read(32 * m), dtype="<u8").reshape(m, 4))
        lag = [poly(ne) for _ in range(3)]
        assert all(m == ne for m, _ in lag)
        assert int.from_bytes(f.read(4), "big") == n_fixed
        values = [poly(n) for _ in range(n_fixed)]
        one = O.fr_from_ints([1])[0]
        table = O.fr_to_ints(values[0][1])
        assert [int(v) for v in table[: 1 << hp.L]] == list(range(1 << hp.L)) and not any(table[1 << hp.L:])      # the lookup table first
        for bits, (_, v) in zip(selectors, values[2:]):
            assert np.array_equal(v[bits], np.broadcast_to(one, v[bits].shape)) and not v[~bits].any()
        assert int.from_bytes(f.read(4), "big") == n_fixed
        coeffs = [poly(n) for _ in range(n_fixed)]
        assert np.array_equal(O.ntt(coeffs[1][1].astype(np.uint64), O.root_of_unity(pr.k)), values[1][1])                # constants: coefficients <-> values
        assert int.from_bytes(f.read(4), "big") == n_fixed
        cosets = [poly(ne) for _ in range(n_fixed)]
        assert np.array_equal(O.coeff_to_extended(coeffs[1][1].astype(np.uint64), 2), cosets[1][1])
    size = len(raw) + 3 * (4 + 32 * ne) + (n_fixed + pr.n_perm) * ((4 + 32 * n) * 2 + 4 + 32 * ne) + 6 * 4
    assert os.path.getsize(pk_path) == size
    want = pr.prove(None, seed=77)
    pr2 = ProverRounds(hp).load_proving_key_raw(pk_path)
    try:
        assert all(np.array_equal(pr2.fixed[name].commits, pr.fixed[name].commits) for name in FIXED)
        assert pr2.instance_cells == pr.instance_cells
        got = pr2.prove(None, seed=77)["proof"]
        assert got == want["proof"] and len(got) > 0
    finally:
        pr2.free()
    write_snark(str(tmp_path / "kmeans.snark"), want["proof"], want["instances"])
    rep = verify_file.main(str(tmp_path / "kmeans.snark"), vk_path, TAU)
    assert rep["accepted"] and not rep["tampered_byte_accepted"]
    assert not verify_file.main(str(tmp_path / "kmeans.snark"), vk_path)["accepted"]             # the reference's SRS is not this fixture's
    other = KmeansHotPath(n=8, dim=4, K=2, I=1, k=12, L=10, metric="cosine", tau=TAU).setup()
    try:
        with pytest.raises(ValueError):
            ProverRounds(other).load_proving_key_raw(pk_path)
    finally:
        other.free()
    with open(pk_path, "r+b") as f:                                         # a damaged file: l_active_row is not this circuit's
        f.seek(len(raw) + 2 * (4 + 32 * ne) + 4 + 5)
        f.write(b"\x01")
    with pytest.raises(ValueError):
        ProverRounds(hp).load_proving_key_raw(pk_path)


def _verify(O, api, proof, vk):
    """A verifier for the proofs ProverRounds.prove(None) writes (SHPLONK): knows the proof bytes and a verifying key — the
    circuit's shape, the fixed commitments, [tau] H — and nothing else.  Replays the transcript, checks the quotient identity
    and the one pairing equation.  Returns True / False (malformed points or scalars: False)."""
    from oracle import pairing as PR
    R, meta, opened = O.R_MOD, vk["meta"], vk["opened"]
    counts = {"adv": meta["n_cols"], "advg": meta["n_adv"], "sel": meta["n_adv"], "sigma": meta["n_cols"] + 2, "cst": 1, "table": 1, "pa": meta["n_lk"], "ps": meta["n_lk"],
              "zp": meta["n_sets"], "zl": meta["n_lk"], "rand": 1, "hf": 0}
    pos = 0
    tr = api.Transcript()
    try:
        def points(m):
            nonlocal pos
            pts = []
            for _ in range(m):
                if pos + 32 > len(proof):
                    raise ValueError
                try:
                    pts.append(_decompress(O, proof[pos: pos + 32]))
                except AssertionError:
                    raise ValueError
                pos += 32
                tr.common_point(pts[-1])
            return np.stack(pts) if pts else np.zeros((0, 8), dtype=np.uint64)
        tr.common_scalar(_vk_digest(api, vk["fixed"]))
        for value in vk.get("instances", []):
            tr.common_scalar(O.fr_from_ints([value])[0])
        C = dict(vk["fixed"])
        C["adv"] = points(counts["adv"])
        C["advg"] = C["adv"][: meta["n_adv"]]              # the gate columns, opened at rows 1..3 as a group of their own
        ch = {"theta": tr.squeeze()}
        pairs = points(2 * meta["n_lk"])
        C["pa"], C["ps"] = pairs[0::2], pairs[1::2]
        ch["beta"], ch["gamma"] = tr.squeeze(), tr.squeeze()
        C["zp"], C["zl"] = points(counts["zp"]), points(counts["zl"])
        C["rand"] = points(1)
        ch["y"] = tr.squeeze()
        C["h"] = points(meta["chunk_len"] + 1)              # the quotient's degree - 1 pieces (chunk_len = degree - 2)
        ch["x"] = tr.squeeze()
        evals = {}
        for rot, names in opened.items():
            for name in names:
                vals = []
                for _ in range(counts[name]):
                    e = int.from_bytes(proof[pos: pos + 32], "little")
                    if pos + 32 > len(proof) or e >= R:
                        raise ValueError
                    pos += 32
                    tr.common_scalar(O.fr_from_ints([e])[0])
                    vals.append(e)
                evals[(name, rot)] = vals
        ch["yo"], ch["v"] = tr.squeeze(), tr.squeeze()
        W1 = points(1)[0]
        ch["u"] = tr.squeeze()
        W2 = points(1)[0]
        if pos != len(proof):
            raise ValueError
    except ValueError:
        return False
    finally:
        tr.free()
    # h folded at x is not in the proof: its value is what the quotient identity demands, its commitment the pieces' combination;
    # the pairing equation below then holds only if the committed h really takes that value
    hf = folded_h(O, meta, ch, evals, vk.get("instances", []), C["h"])
    if hf is None:
        return False
    evals[("hf", 0)], C["hf"] = [hf[0]], hf[1].reshape(1, 8)
    to_int = lambda a: O.fr_to_ints(np.asarray(a).reshape(1, 4))[0]
    x, yo, v, u = (to_int(ch[n]) for n in ("x", "yo", "v", "u"))
    w = to_int(O.root_of_unity(meta["k"]))
    pts = {rot: x * pow(w, rot % meta["rows"], R) % R for rot in opened}
    by_poly = {}
    for rot, names in opened.items():
        for name in names:
            by_poly.setdefault(name, []).append(rot)
    sets = []
    for name, rots in by_poly.items():
        key = tuple(sorted(rots))
        for sset in sets:
            if sset[0] == key:
                sset[1].append(name)
                break
        else:
            sets.append((key, [name]))
    all_rots = sorted({rot for rots, _ in sets for rot in rots})
    def vanish(rots, at):
        acc = 1
        for rot in rots:
            acc = acc * (at - pts[rot]) % R
        return acc
    m, scalars, bases, g_scalar = len(sets), [], [], 0
    for s_i, (rots, names) in enumerate(sets):
        commits = np.concatenate([C[name] for name in names])
        vals = []
        for rot in rots:
            acc = 0
            for name in names:
                for e in evals[(name, rot)]:
                    acc = (acc * yo + e) % R
            vals.append(acc)
        r_u = 0
        for c in reversed(_interpolate([pts[rot] for rot in rots], vals, R)):
            r_u = (r_u * u + c) % R
        coef = pow(v, m - 1 - s_i, R) * vanish([rot for rot in all_rots if rot not in rots], u) % R
        scalars += [coef * pow(yo, commits.shape[0] - 1 - i, R) % R for i in range(commits.shape[0])]
        bases.append(commits)
        g_scalar = (g_scalar - coef * r_u) % R
    scalars += [g_scalar, (-vanish(all_rots, u)) % R, u]
    bases += [O.g1_generator().reshape(1, 8), W1.reshape(1, 8), W2.reshape(1, 8)]
    # (double-and-add per point for the circuits of the tests; the bucket method above a few thousand points: C4' combines 58 k commitments)
    combine = O.msm_naive if len(scalars) < 4096 else (lambda sc, pts: O.msm(sc, pts, threads=8))
    left = combine(O.fr_from_ints(scalars), np.concatenate(bases))
    to_pt = lambda a: None if not np.asarray(a).any() else tuple(O.fq_to_ints(np.asarray(a).reshape(2, 4)))
    return PR.pairing_product_is_one([(to_pt(left), PR.G2), (PR.pt_neg(to_pt(W2)), vk["tau_h"])])


def test_a_verifier_accepts_the_proof_bytes_and_rejects_tampered_ones(circuit, O):
    from halo2_vectordb_amd import api
    from oracle import pairing as PR
    hp, pr = circuit
    out = pr.prove(None, seed=31)
    vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED},
              tau_h=PR.pt_mul(PR.G2, TAU), instances=out["instances"])
    proof = out["proof"]
    assert _verify(O, api, proof, vk)
    assert not _verify(O, api, proof, {**vk, "instances": []})       # a proof is a proof of its statement: the public centroids
    # through the proof file the Prove arm leaves behind (io.write_snark / read_snark)
    from halo2_vectordb_amd.io import read_snark, write_snark
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        write_snark(os.path.join(d, "kmeans.snark"), proof, out["instances"])
        proof_f, inst_f = read_snark(os.path.join(d, "kmeans.snark"))
        # ... and the verifying-key file of the Keygen arm (ProverRounds.save_verifying_key / io.read_verifying_key): the pair of files
        # is all the stand-alone verifier (tests/verify_file.py) is given
        pr.save_verifying_key(os.path.join(d, "kmeans.snark.vk.npz"), opened=out["opened"])
        import verify_file
        rep = verify_file.main(os.path.join(d, "kmeans.snark"))
        assert rep["accepted"] and not rep["tampered_byte_accepted"] and rep["columns"] == pr.n_cols
    assert proof_f == proof and _verify(O, api, proof_f, {**vk, "instances": inst_f})
    n_h = vk["meta"]["chunk_len"] + 1
    n_points = vk["meta"]["n_cols"] + 3 * vk["meta"]["n_lk"] + vk["meta"]["n_sets"] + 1 + n_h
    # a commitment, the random polynomial, h, an evaluation, the last evaluation (the random polynomial's), W1, W2
    for where in (5, 32 * (n_points - n_h - 1) + 3, 32 * (n_points - 1) + 3, 32 * n_points + 40, len(proof) - 96 + 9, len(proof) - 64 + 7, len(proof) - 20):
        bad = bytearray(proof)
        bad[where] ^= 4
        assert not _verify(O, api, bytes(bad), vk)
    assert not _verify(O, api, proof[:-32], vk) and not _verify(O, api, proof + bytes(32), vk)
    # a verifying key that states another constraint degree than the circuit's (chunk_len = degree - 2 columns per product polynomial,
    # degree - 1 quotient pieces; [UPSTREAM-RECALL] for halo2's value, rounds.constraint_degree): the proof is not a proof under it
    for wrong in (1, 3):
        n_sets = -(-(vk["meta"]["n_cols"] + 2) // wrong)
        assert not _verify(O, api, proof, {**vk, "meta": {**vk["meta"], "chunk_len": wrong}})
        assert not _verify(O, api, proof, {**vk, "meta": {**vk["meta"], "chunk_len": wrong, "n_sets": n_sets}})
    assert not _verify(O, api, proof, {**vk, "meta": {**vk["meta"], "n_blind": vk["meta"]["n_blind"] - 1}})


def test_every_proof_draws_fresh_blinding_scalars(circuit, O):
    """halo2's create_proof blinds every committed polynomial with Scalar::random(OsRng): two proofs of the same witness
    must not share a commitment (shared blinds would expose the difference of two witnesses), and both must verify"""
    from halo2_vectordb_amd import api
    from oracle import pairing as PR
    hp, pr = circuit
    a, b = pr.prove(None), pr.prove(None)               # seed=None: operating-system entropy
    for name in ("adv", "pa", "ps", "zp", "zl", "rand", "h"):
        ca, cb = a["commitments"][name], b["commitments"][name]
        assert ca.shape == cb.shape and not (ca == cb).all(axis=1).any(), name
    assert a["proof"] != b["proof"]
    for out in (a, b):
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU), instances=out["instances"])
        assert _verify(O, api, out["proof"], vk)
    # the hot path alone: two steps, different advice commitments for the same witness; a seed (test hook) pins them
    c1, c2 = hp.step().copy(), hp.step().copy()
    assert not (c1 == c2).all(axis=1).any()
    assert np.array_equal(hp.step(blind_seed=5), hp.step(blind_seed=5))
    # the blinding rows are uniform field elements: canonical values below r that fill the whole range
    blinds = O.fr_to_ints(hp.d_blind.download((hp.n_cols * 7, 4)))
    assert len(set(blinds)) == len(blinds) and max(v.bit_length() for v in blinds) >= 252 and all(v < O.R_MOD for v in blinds)


def test_merkle_copy_map_matches_the_witness_and_closes_the_permutation(O):
    """The Merkle circuit's own copy constraints (halo2_vectordb_amd/copymap.py: symbolic trace of PoseidonChip::permutation,
    checked cell by cell against the kernel's gate / constant flags): on a real witness every cell equals the cell it copies
    — two different databases —, most cells are tied, the root cell holds the root; with the whole map in the permutation
    argument the proof still verifies (the products close), and the stand-alone verifier accepts it."""
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.copymap import mapping_from_copy_of
    from halo2_vectordb_amd.pipeline import MerkleHotPath
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    from oracle import pairing as PR
    hp = MerkleHotPath(n=6, dim=5, k=11, tau=TAU).setup()      # 6 leaves padded to 8 (the zero cell), odd width: 3 permutations per leaf
    pr = ProverRounds(hp).keygen()
    try:
        cm = pr.circuit
        copy_of = cm.copy_of
        assert copy_of.size == hp.n_cells and (copy_of <= np.arange(copy_of.size)).all()
        assert pr.keygen_report.violations() == 0, pr.keygen_report.as_dict()
        flags_d = hp.keygen_flags()
        flagged = (flags_d.download((hp.n_cells,), dtype=np.uint8) & 2) != 0
        flags_d.free()
        # the constants: every cell the kernels flag, and the initial sponge state of 6 leaves and 7 tree nodes they emit unflagged
        assert (cm.const_idx[flagged] >= 0).all() and int(((cm.const_idx >= 0) & ~flagged).sum()) == 3 * (6 + 7)
        assert (1 << 64) in cm.consts and 0 in cm.consts and 1 in cm.consts
        tied = int((copy_of != np.arange(copy_of.size)).sum())
        assert tied > 0.4 * copy_of.size
        for seed in (1, 2):
            vec = np.random.default_rng(seed).integers(0, 219, size=(6, 5)).astype(np.float64)
            hp.set_vectors(vec)
            hp._witness()
            api.sync()
            stream = hp.d_stream.download((hp.n_cells, 4))
            assert np.array_equal(stream, stream[copy_of])
            assert np.array_equal(stream[: 6 * 5], hp.qvec.reshape(-1, 4))                  # the assigned vector words ...
            assert (copy_of[30:] < 30).sum() == 6 * 5                                       # ... each absorbed exactly once
            assert np.array_equal(stream[pr.root_cell], api.poseidon_merkle_root(hp.qvec))
            cst = np.flatnonzero(cm.const_idx >= 0)                                         # every constant cell holds its constant
            assert np.array_equal(stream[cst], O.fr_from_ints(cm.consts)[cm.const_idx[cst]])
        # the permutation built from it is a permutation of [advice | constants | instance]; the root is the one public cell
        assert pr.instance_cells == [pr.root_cell]
        mapping = mapping_from_copy_of(copy_of, hp.bp, pr.n_cols, pr.rows, const_idx=cm.const_idx, n_consts=len(cm.consts), instance_cells=pr.instance_cells)
        flat = (mapping >> np.uint64(32)).astype(np.int64) * pr.rows + (mapping & np.uint64(0xFFFFFFFF)).astype(np.int64)
        assert mapping.shape[0] == pr.n_cols + 2 and np.array_equal(np.sort(flat.reshape(-1)), np.arange((pr.n_cols + 2) * pr.rows))
        out = pr.prove(None, seed=8)
        # no lookup columns, so no lookup argument: cs.degree() = 3 — one column per permutation product, two quotient pieces, and the
        # extended domain of 2 n points (the two cosets the rounds evaluate on are all of it)
        assert pr.n_lk == 0 and pr.degree == 3 and pr.chunk_len == 1 and pr.n_sets == pr.n_cols + 2 and pr.n_slots == 2
        assert out["commitments"]["h"].shape == (2, 8) and out["commitments"]["zp"].shape == (pr.n_cols + 2, 8)
        assert quotient_identity_holds(pr, out["challenges"], out["evals"], out["instances"])
        assert out["instances"] == [O.fr_to_ints(api.poseidon_merkle_root(hp.qvec).reshape(1, 4))[0]]                # the public input is the root
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED},
                  tau_h=PR.pt_mul(PR.G2, TAU), instances=out["instances"])
        assert _verify(O, api, out["proof"], vk)
    finally:
        pr.free()
        hp.free()


def test_copy_constraints_are_enforced(O):
    """A vector word at the head of the stream is in no gate: changing it after witness generation leaves every gate and every
    constant intact and breaks exactly one copy constraint (the word a leaf absorbs is no longer the assigned one).  With the
    Merkle circuit's copy map in the permutation argument the verifier rejects the proof; with only the layout's own ties
    (the identity map) the same tampered witness still proves — which is what the map is for."""
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.pipeline import MerkleHotPath
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    from oracle import pairing as PR
    hp = MerkleHotPath(n=4, dim=4, k=11, tau=TAU).setup()
    honest = hp._witness
    other = O.fr_from_ints([123456789])

    def tampered(sel=None):
        honest(sel)
        if sel is None:
            hp.d_stream.upload(other, offset=7 * 32)

    tau_h = PR.pt_mul(PR.G2, TAU)
    results = {}
    from halo2_vectordb_amd.circuit_sym import CopyMap
    for label in ("full map", "layout ties only"):
        pr = ProverRounds(hp).keygen()
        if label == "layout ties only":        # the same circuit with the copies of the Poseidon trace left out (constants stay pinned)
            cm = pr.circuit
            loose = CopyMap(np.arange(hp.n_cells, dtype=np.int64), cm.const_idx.copy(), list(cm.consts), cm.asserted.copy(), cm.gate.copy(), cm.lookup_src.copy())
            pr.free()                            # (releases the map's device arrays too)
            pr = ProverRounds(hp).keygen(circuit=loose)
        try:
            vk = lambda out: dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=tau_h,
                                  instances=out["instances"])
            out = pr.prove(None, seed=1)
            assert _verify(O, api, out["proof"], vk(out))                          # the honest witness proves under both
            hp._witness = tampered
            try:
                bad = pr.prove(None, seed=1)
            finally:
                hp._witness = honest
            results[label] = (quotient_identity_holds(pr, bad["challenges"], bad["evals"], bad["instances"]), _verify(O, api, bad["proof"], vk(bad)))
        finally:
            pr.free()
    hp.free()
    assert results["full map"] == (False, False)
    assert results["layout ties only"] == (True, True)


def test_lookup_cells_are_tied_to_the_advice_cells_they_copy(circuit, O):
    """cells_to_lookup holds copies of advice cells: keygen marks the sources (flag bit 2) and ties every lookup cell to its
    source through the permutation argument.  On a real k-means witness every lookup cell equals its source; and a witness
    whose lookup column is given a different (still in-table) value no longer proves, whereas it does when the ties are left out
    — a range check that is not tied to the cell it checks proves nothing about it."""
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.rounds import ProverRounds, quotient_identity_holds
    hp, pr = circuit
    from halo2_vectordb_amd.circuit_sym import CopyMap
    from halo2_vectordb_amd.copymap import lookup_sources
    cm = pr.circuit
    assert len(cm.lookup_src) == hp.n_lookup
    hp._witness()
    api.sync()
    stream, lookup = hp.d_stream.download((hp.n_cells, 4)), hp.d_lookup.download((hp.n_lookup, 4))
    assert np.array_equal(lookup, stream[cm.lookup_src])
    # the kernels mark the same cells as lookup sources (flag bit 2) in the same order
    d_flags = hp.keygen_flags()
    assert np.array_equal(lookup_sources(d_flags.download((hp.n_cells,), dtype=np.uint8), hp.n_lookup), cm.lookup_src)
    d_flags.free()
    honest = hp._witness
    swapped = O.fr_from_ints([(O.fr_to_ints(lookup[5].reshape(1, 4))[0] + 1) % (1 << hp.L)])       # another value of the table

    def tampered(sel=None):
        honest(sel)
        if sel is None:
            hp.d_lookup.upload(swapped, offset=5 * 32)

    loose = ProverRounds(hp).keygen(circuit=CopyMap(cm.copy_of, cm.const_idx, cm.consts, cm.asserted, cm.gate, None))
    results = {}
    try:
        for label, p in (("tied", pr), ("untied", loose)):
            hp._witness = tampered
            try:
                bad = p.prove(None, seed=2)
            finally:
                hp._witness = honest
            results[label] = quotient_identity_holds(p, bad["challenges"], bad["evals"], bad["instances"])
    finally:
        loose.free()
    assert results == {"tied": False, "untied": True}


def test_rounds_on_a_nearest_vector_circuit(O):
    """the same rounds, copy ties (layout + lookup sources) and verifier on nearest_vector (query + 12 vectors of 6 words)"""
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd.pipeline import NearestHotPath
    from halo2_vectordb_amd.rounds import ProverRounds
    from oracle import pairing as PR
    hp = NearestHotPath(n=12, dim=6, k=12, L=11, tau=TAU).setup()
    pr = ProverRounds(hp).keygen()
    try:
        assert pr.n_lk >= 1 and len(pr.circuit.lookup_src) == hp.n_lookup and pr.keygen_report.violations() == 0
        out = pr.prove(None, seed=6)
        vk = dict(meta=_meta(pr), opened=out["opened"], fixed={name: pr.fixed[name].commits for name in FIXED}, tau_h=PR.pt_mul(PR.G2, TAU),
                  instances=out["instances"])
        assert _verify(O, api, out["proof"], vk)
    finally:
        pr.free()
        hp.free()


def test_sigma_columns_from_the_packed_mapping(circuit, O):
    """The product round reads the sigma columns in Lagrange form: made from the 32-bit packed mapping kept with the key (one product per
    cell), they equal the forward transform of the coefficient form the key holds, and a proof made with a key that does not hold the
    mapping (the file-loaded case: transform route) has the same bytes."""
    import ctypes
    from halo2_vectordb_amd import api
    from halo2_vectordb_amd._lib import check
    hp, pr = circuit
    lib, rows, k, B = pr.lib, pr.rows, pr.k, 32
    assert pr.d_map32 is not None
    c0, nb = 1, min(4, pr.n_perm - 1)
    d_a, d_b = api.DeviceBuffer(nb * rows * B), api.DeviceBuffer(nb * rows * B)
    check(lib.vdb_permutation_sigma_packed_dev(pr.d_map32.at(c0 * rows * 4), ctypes.c_size_t(nb), ctypes.c_size_t(pr.n_perm), k, api._p(pr.delta), d_a.ptr))
    check(lib.vdb_memcpy_d2d(d_b.ptr, pr.fixed["sigma"].coeff.at(c0 * rows * B), ctypes.c_size_t(nb * rows * B)))
    check(lib.vdb_ntt_batch_dev(d_b.ptr, ctypes.c_size_t(nb), k, api._p(api.root_of_unity(k)), 0))
    api.sync()
    assert np.array_equal(d_a.download((nb, rows, 4)), d_b.download((nb, rows, 4)))
    for d in (d_a, d_b):
        d.free()
    with_map = pr.prove(None, seed=77)["proof"]
    held, pr.d_map32 = pr.d_map32, None
    try:
        without = pr.prove(None, seed=77)["proof"]
    finally:
        pr.d_map32 = held
    assert with_map == without
    check(lib.vdb_permutation_sigma_packed_dev(pr.d_map32.ptr, ctypes.c_size_t(0), ctypes.c_size_t(pr.n_perm), k, api._p(pr.delta), pr.d_map32.ptr))   # empty block
    assert lib.vdb_permutation_sigma_packed_dev(pr.d_map32.ptr, ctypes.c_size_t(pr.n_perm + 1), ctypes.c_size_t(pr.n_perm), k, api._p(pr.delta), pr.d_map32.ptr) == -3
    # the untimed proof evaluates and absorbs group by group (device evaluation beside host absorption), the instrumented one evaluates
    # everything first: the same bytes
    assert pr.prove(None, seed=77, timings={})["proof"] == with_map
    # a mapping whose columns and rows do not fit 32 bits together is refused
    d = api.DeviceBuffer(64)
    assert lib.vdb_permutation_mapping_pack_dev(d.ptr, ctypes.c_size_t(1 << 20), 16, d.ptr) == -3
    d.free()
