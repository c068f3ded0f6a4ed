"""Runs of equal coefficients in commitPoly (round 5): the running sums of the alpha basis the SRS handle holds, and the unprepared S_j
of prove() that is committed through them -- c (A[a] + ... + A[b]) = c ps[b] - c ps[a - 1] -- against the CPU oracle, bit-exact."""
import os
import random

import numpy as np
import pytest

from util import NCPU, R, big_circuit, circuit_arrays, fr_bytes, rand_fr_array

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _runs_at_any_size():
    """by default prove() takes the runs out from n = 2^16 (below, the extra launches cost more than they save); SONIC_PROVE_RUNS=1 takes
    them out wherever there are 8 tiles, so that seconds-sized cases reach the path.  Read at every proof."""
    os.environ["SONIC_PROVE_RUNS"] = "1"
    os.environ["SONIC_PROVE_SYM"] = "1"              # (and C over the SRS's symmetric sums: same default rule, same switch values)
    yield
    os.environ.pop("SONIC_PROVE_RUNS", None)
    os.environ.pop("SONIC_PROVE_SYM", None)


PREFIX = 1000          # SONIC_BASIS_ALPHA_PREFIX (include/sonic_hip.h)
SYM = 1001             # SONIC_BASIS_ALPHA_SYM
INF = bytes(96)


def _prefix_by_oracle(orc, pts):
    acc, out = INF, []
    for p in pts:
        b = p.tobytes()
        acc = b if acc == INF else (acc if b == INF else orc.g1_add(acc, b))
        out.append(acc)
    return out


@pytest.mark.parametrize("d,x", [(100, None), (40, 1), (3, 5)])
def test_running_sums_small(sonic, orc, d, x):
    """every entry of the table against sequential oracle additions; x = 1 makes all points of the basis equal (the scan's additions
    are doublings: the exceptional branch of the general addition)"""
    pyr = random.Random(d)
    s = sonic.SRS.new(d, x or pyr.randrange(2, R), pyr.randrange(2, R))
    n = 2 * d + 1
    pts = s.points(1, -d, n)
    assert pts[d].tobytes() == INF                                    # the omitted g^alpha adds nothing
    want = _prefix_by_oracle(orc, pts)
    got = s.points(PREFIX, -d, n)
    for i in range(n):
        assert got[i].tobytes() == want[i], i
    s.close()


@pytest.mark.parametrize("d,x", [(100, None), (40, 1)])
def test_symmetric_sums_small(sonic, orc, d, x):
    """sym[e] = A[e] + A[-e] for e in [1, d], empty below; x = 1 makes them doublings"""
    pyr = random.Random(d + 7)
    s = sonic.SRS.new(d, x or pyr.randrange(2, R), pyr.randrange(2, R))
    pts = s.points(1, -d, 2 * d + 1)
    got = s.points(SYM, -d, 2 * d + 1)
    for e in range(-d, d + 1):
        want = orc.g1_add(pts[d + e].tobytes(), pts[d - e].tobytes()) if e >= 1 else INF
        assert got[d + e].tobytes() == want, e
    s.close()


def test_symmetric_sums_tables_at_scale(sonic, orc):
    """d = 2^20 + 3 (two slabs of the builder, the second one short): an MSM over the symmetric sums' WINDOW TABLES -- reached through prove()'s
    C only, so checked here through a proof: n = 2^17 with the commitments of the oracle -- is covered by test_runs_by_default_at_2p16 and the
    full-size byte tests; here the table itself at slab borders"""
    d = (1 << 20) + 3
    s = sonic.SRS.new(d, 0x1234567, 0x7654321)
    for e in (1, 2, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, d):
        a, b = s.points(1, e, 1)[0].tobytes(), s.points(1, -e, 1)[0].tobytes()
        assert s.points(SYM, e, 1)[0].tobytes() == orc.g1_add(a, b), e
    assert s.points(SYM, 0, 1)[0].tobytes() == INF and s.points(SYM, -5, 1)[0].tobytes() == INF
    s.close()


def test_running_sums_across_slabs(sonic, orc):
    """d = 2^20: 2^21 + 1 points, three slabs of the builder; entries at chunk, slab and table borders against the MSM entry point with
    all-one scalars over the same prefix (the table-driven Pippenger: other code), and one slab border against the oracle's Pippenger"""
    from sonic_amd.commitment import msm_g1_srs
    d = 1 << 20
    s = sonic.SRS.new(d, 0x1234567, 0x7654321)
    n = 2 * d + 1
    one = np.zeros((n, 32), np.uint8)
    one[:, 0] = 1
    for m in (1, 15, 16, 17, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, (1 << 21) - 1, 1 << 21, n):
        sc = one[:m].copy()
        if m > d:
            sc[d, 0] = 0                                                # (the omitted slot: a non-zero coefficient there is an index error)
        assert s.points(PREFIX, -d + m - 1, 1)[0].tobytes() == msm_g1_srs(s, 1, -d, sc), m
    m = (1 << 20) + 16
    assert s.points(PREFIX, -d + m - 1, 1)[0].tobytes() == orc.msm(s.points(1, -d, m)[np.arange(m) != d], np.ascontiguousarray(one[:m - 1]), 1, NCPU)
    s.close()


def _prove_all_ways(sonic, srs, circ_enc, tr, n, Q):
    """the proof bytes of one statement: prepared handle, plain handle (runs through the running sums), and the same with the runs switched
    off -- all must agree"""
    circuit = sonic.ArithCircuit(sonic.GateWeights(circ_enc["wL"], circ_enc["wR"], circ_enc["wO"]), circ_enc["cs"])
    asg = sonic.Assignment(circ_enc["aL"], circ_enc["aR"], circ_enc["aO"])
    out = []
    for prepare in (False, True):
        p = sonic.Prover(srs, circuit, prepare=prepare)
        p.set_assignment(asg)
        out.append(p.prove_bytes(tr))
        out.append(p.prove_bytes(tr))                                  # second proof through the same handle (buffers re-used)
        p.set_share(0, 1)                                              # the share path with a world of one: the whole proof as one share
        out.append(sonic.proof_from_shares(Q, [p.prove_share(tr)], tr))
        p.close()
    return out


@pytest.mark.parametrize("n,Q,shape", [(700, 2, "rows"), (1000, 3, "rows"), (4096, 2, "rows"), (5000, 1, "rows"), (3000, 2, "runs"), (2048, 4, "dense")])
def test_prove_with_runs_matches_oracle(sonic, orc, n, Q, shape):
    """prove() on a handle that is not prepared commits S_j with the runs taken out; circuits: the reference's rndCircuit rows (one
    all-ones row per matrix: n equal coefficients twice, test/Test/Reference.hs:141-155), rows made of several runs of different values
    with borders off the tile grid and single odd values inside, and dense random rows (no run at all)"""
    d = 8 * n
    pyr = random.Random(n + Q)
    srs = sonic.SRS.new(d, pyr.randrange(2, R), pyr.randrange(2, R))
    osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1))
    c = big_circuit(n * 7 + Q, n, Q)
    if shape != "rows":
        rng = np.random.default_rng(n)
        la, lb, lo = c["ints"]
        rows = []
        for _ in range(3):
            w = rand_fr_array(rng, Q * n).reshape(Q, n, 32)
            if shape == "runs":
                for q in range(Q):
                    pos = 0
                    while pos < n:
                        ln = int(rng.integers(1, 900))
                        if rng.integers(0, 3):
                            w[q, pos:pos + ln] = rand_fr_array(rng, 1)[0]
                        pos += ln
                    w[q, n // 2] = rand_fr_array(rng, 1)[0]             # one odd value inside whatever run is there
            rows.append(w)
        ints = [[[int.from_bytes(w[q, i].tobytes(), "little") for i in range(n)] for q in range(Q)] for w in rows]
        cs = [(sum(a * b for a, b in zip(ints[0][q], la)) + sum(a * b for a, b in zip(ints[1][q], lb)) + sum(a * b for a, b in zip(ints[2][q], lo))) % R
              for q in range(Q)]
        c = dict(c, wL=rows[0].reshape(-1, 32), wR=rows[1].reshape(-1, 32), wO=rows[2].reshape(-1, 32), cs=fr_bytes(cs))
    tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
    want = orc.prove(osrs, n, Q, c["wL"], c["wR"], c["wO"], c["cs"], c["aL"], c["aR"], c["aO"], tr)
    got = _prove_all_ways(sonic, srs, c, tr, n, Q)
    assert all(g == want for g in got), [g == want for g in got]
    proof, _ = sonic.prove(srs, sonic.Assignment(c["aL"], c["aR"], c["aO"]),
                           sonic.ArithCircuit(sonic.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"]),
                           transcript=[int.from_bytes(tr[i].tobytes(), "little") for i in range(8 + 2 * Q)])
    assert proof.to_bytes() == want                                     # the one-shot call (a shell that is never prepared)
    srs.close()


def test_runs_knobs_give_the_same_bytes(sonic):
    """SONIC_SRS_PREFIX=0 / SONIC_SRS_SYM=0: an SRS without the running / symmetric sums (its S_j and C are plain MSMs); SONIC_PROVE_RUNS=0 /
    SONIC_PROVE_SYM=0: the handle ignores them; unset: the default rule (this n is below it: plain).  All give the same proof."""
    n, Q = 3000, 2
    d = 8 * n
    c = big_circuit(5, n, Q)
    tr = fr_bytes([random.Random(1).randrange(1, R) for _ in range(8 + 2 * Q)])
    os.environ["SONIC_SRS_PREFIX"] = "0"
    os.environ["SONIC_SRS_SYM"] = "0"
    try:
        plain = sonic.SRS.new(d, 77, 99)
    finally:
        del os.environ["SONIC_SRS_PREFIX"]
        del os.environ["SONIC_SRS_SYM"]
    with pytest.raises(sonic.SonicError):
        plain.points(PREFIX, 0, 1)
    with pytest.raises(sonic.SonicError):
        plain.points(SYM, 1, 1)
    full = sonic.SRS.new(d, 77, 99)
    a = _prove_all_ways(sonic, plain, c, tr, n, Q)
    b = _prove_all_ways(sonic, full, c, tr, n, Q)
    os.environ["SONIC_PROVE_RUNS"] = "0"
    os.environ["SONIC_PROVE_SYM"] = "0"
    b += _prove_all_ways(sonic, full, c, tr, n, Q)
    os.environ["SONIC_PROVE_SYM"] = "1"
    b += _prove_all_ways(sonic, full, c, tr, n, Q)                      # C over the symmetric sums, S_j plain
    del os.environ["SONIC_PROVE_RUNS"]
    del os.environ["SONIC_PROVE_SYM"]
    b += _prove_all_ways(sonic, full, c, tr, n, Q)
    assert len(set(a + b)) == 1
    plain.close(); full.close()


def test_runs_by_default_at_2p16(sonic, orc):
    """the default rule: at n = 2^16 a handle that is not prepared takes the runs out without being asked to; bytes against the oracle"""
    os.environ.pop("SONIC_PROVE_RUNS", None)
    os.environ.pop("SONIC_PROVE_SYM", None)
    n, Q = 1 << 16, 2
    d = 8 * n
    srs = sonic.SRS.new(d, 12345, 67890)
    osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1))
    c = big_circuit(16, n, Q)
    tr = fr_bytes([random.Random(16).randrange(1, R) for _ in range(8 + 2 * Q)])
    want = orc.prove(osrs, n, Q, c["wL"], c["wR"], c["wO"], c["cs"], c["aL"], c["aR"], c["aO"], tr)
    assert all(g == want for g in _prove_all_ways(sonic, srs, c, tr, n, Q))
    srs.close()
