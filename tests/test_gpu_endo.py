"""The endomorphism form of the fixed-base MSM (sonic_amd/csrc/endo.hpp): an SRS that holds 7 window tables over 130 bits instead of
13 over 255 -- what `SRS.new` falls back to when the full tables do not fit in HBM (d >= 2^25), forced here with SONIC_MSM_ENDO=1 at
sizes the oracle can follow.  Every scalar is split s = s1 + lambda s2 on the device, the two half-scalar MSMs run over the same
points, and the host adds phi(second sum): results must be the same group elements -- the same bytes -- as over the full tables
and as the CPU oracle's (the folds at src/Sonic/CommitmentScheme.hs:25-29, 45-48)."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from util import NCPU, R, big_circuit, circuit_arrays, fr_bytes, rand_fr_array

pytestmark = pytest.mark.gpu
LAM = 0xd201000000010000 ** 2 - 1


@pytest.fixture()
def endo_env(monkeypatch):
    monkeypatch.setenv("SONIC_MSM_ENDO", "1")
    yield
    monkeypatch.delenv("SONIC_MSM_ENDO", raising=False)


def _plan(sonic, srs, n):
    from sonic_amd import _lib
    c, w, sets = C.c_int(), C.c_int(), C.c_int()
    _lib.check(_lib.lib().sonic_msm_plan(srs._h, n, C.byref(c), C.byref(w), C.byref(sets)))
    return c.value, w.value, sets.value


def test_endo_msm_matches_full_tables_and_oracle(sonic, orc, endo_env):
    from sonic_amd.commitment import msm_g1_srs
    d, x, alpha = 1 << 13, 0x1234567891, 0x9876543211
    srs_e = sonic.SRS.new(d, x, alpha)
    os.environ["SONIC_MSM_ENDO"] = "0"
    srs_f = sonic.SRS.new(d, x, alpha)
    os.environ["SONIC_MSM_ENDO"] = "1"
    osrs = orc.SRS(d, x, alpha, threads=NCPU)
    ce, we, se = _plan(sonic, srs_e, 5000)
    cf, wf, sf = _plan(sonic, srs_f, 5000)
    assert se == sf == 1 and wf == (255 + cf - 1) // cf and we == (130 + ce - 1) // ce and we < wf and we * 2 <= wf + 2
    assert np.array_equal(srs_e.points(0, -d, 2 * d + 1), srs_f.points(0, -d, 2 * d + 1))        # the bases themselves are the same
    pyr = random.Random(3)
    edge = [0, 1, 2, LAM - 1, LAM, LAM + 1, 2 * LAM, R - 1, R - 2, R - LAM, (R - 1) // 2, (R + 1) // 2, LAM * LAM % R, (1 << 128) - 1, 1 << 128, 1 << 254]
    for n in (1, 3, len(edge), 700, 5000, 16000):
        sc = fr_bytes((edge + [pyr.randrange(R) for _ in range(n)])[:n]) if n <= 700 else rand_fr_array(np.random.default_rng(n), n)
        for basis, e0 in ((0, -d), (1, 1 if n < d else -d), (0, d - n + 1)):          # (basis 1 from -d crosses the empty slot e = 0)
            got = msm_g1_srs(srs_e, basis, e0, sc)
            assert got == msm_g1_srs(srs_f, basis, e0, sc) == orc.msm_srs(osrs, basis, e0, sc, 1, NCPU), (n, basis)
    # many equal scalars (heavy buckets) and all-zero scalars
    sc = fr_bytes([pyr.randrange(R)] * 6000 + [0] * 50)
    assert msm_g1_srs(srs_e, 0, -3000, sc) == orc.msm_srs(osrs, 0, -3000, sc, 1, NCPU)
    assert msm_g1_srs(srs_e, 0, 0, fr_bytes([0] * 4000)) == bytes(96)
    # the bucket exchange of a bucket-sharded MSM needs the full tables: refused, not wrong
    from sonic_amd import _lib, distributed as sd
    with pytest.raises(_lib.SonicError):
        sd.exchange_layout(srs_e, 2)


@pytest.mark.parametrize("n,Q,prepare", [(16, 2, False), (300, 3, True), (1 << 12, 2, True)])
def test_endo_prove_bytes(sonic, orc, ref, endo_env, n, Q, prepare):
    """prove() over an endomorphism SRS: the same proof bytes as the oracle's, also as three ranks' shares, and commitPoly / openPoly"""
    pyr = random.Random(n)
    d = 8 * n
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    srs = sonic.SRS.new(d, x, alpha)
    osrs = orc.SRS(d, x, alpha, threads=NCPU)
    if n <= 300:
        circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
        circuit = sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3])
        assignment = sonic.Assignment(*asg)
    else:
        enc = big_circuit(9, n, Q)
        circuit = sonic.ArithCircuit(sonic.GateWeights(enc["wL"], enc["wR"], enc["wO"]), enc["cs"])
        assignment = sonic.Assignment(enc["aL"], enc["aR"], enc["aO"])
    tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
    want = orc.prove(osrs, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(tr))
    p = sonic.Prover(srs, circuit, prepare=prepare)
    p.set_assignment(assignment)
    assert p.prove_bytes(tr) == want
    shares = []
    for r in range(3):
        p.set_share(r, 3)
        shares.append(p.prove_share(tr))
    assert sonic.proof_from_shares(Q, shares, tr) == want
    p.close()
    poly = [(e, pyr.randrange(1, R)) for e in sorted(pyr.sample(range(-d + 1, d), min(200, d))) if e != 0]
    exps = np.array([e for e, _ in poly], np.int64)
    co = fr_bytes([c for _, c in poly])
    assert sonic.g1_to_bytes(sonic.commit_poly(srs, d, poly)) == orc.commit_poly(osrs, d, exps, co)
    z = pyr.randrange(1, R)
    fz, W = sonic.open_poly(srs, z, poly)
    ofz, oW = orc.open_poly(osrs, z, exps, co)
    assert fz == ofz and sonic.g1_to_bytes(W) == oW


def test_endo_prove_at_the_bench_size_equals_full_tables(sonic, endo_env):
    """n = 2^18, d = 2^21 -- where the endomorphism plan has its production shape (7 windows of 19 / 18 bits, 2^18 buckets per half, the
    same tree and sort paths as d >= 2^25): the proof over the endomorphism SRS equals, byte for byte, the proof over the full tables
    (which tests/test_gpu_fullsize.py pins to the oracle at this size), whole and as eight ranks' shares"""
    n, Q = 1 << 18, 2
    d = 8 * n
    x, alpha = 0x1234567891abcdef, 0xfedcba9876543211
    circ = big_circuit(1818, n, Q)
    circuit = sonic.ArithCircuit(sonic.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
    asg = sonic.Assignment(circ["aL"], circ["aR"], circ["aO"])
    tr = rand_fr_array(np.random.default_rng(18), 8 + 2 * Q)
    tr[:, 0] |= 1
    srs_e = sonic.SRS.new(d, x, alpha)
    assert _plan(sonic, srs_e, 3 * n) == (19, 7, 1)
    pe = sonic.Prover(srs_e, circuit)
    pe.set_assignment(asg)
    got = pe.prove_bytes(tr)
    shares = []
    for r in range(8):
        pe.set_share(r, 8)
        shares.append(pe.prove_share(tr))
    assert sonic.proof_from_shares(Q, shares, tr) == got
    pe.close()
    srs_e.close()
    os.environ["SONIC_MSM_ENDO"] = "0"
    try:
        srs_f = sonic.SRS.new(d, x, alpha)
    finally:
        os.environ["SONIC_MSM_ENDO"] = "1"
    assert _plan(sonic, srs_f, 3 * n) == (20, 13, 1)
    pf = sonic.Prover(srs_f, circuit)
    pf.set_assignment(asg)
    assert pf.prove_bytes(tr) == got
    pf.close()
