"""The oracle against itself and against the committed golden vectors (CPU, no GPU):
  * constants (q, r, generator) checked numerically,
  * the reference's own QuickCheck properties restated on the literal python restatement
    (test/Test/Constraints.hs, test/Test/CommitmentScheme.hs, test/Test/Signature.hs,
    test/Test/Protocol.hs) -- pairing checks through the known-trapdoor Fr identity,
  * the plain-C oracle == the python restatement (goldens + fresh random cases),
  * the error contract of prove / commitPoly.
"""
import json
import os
import random

import numpy as np
import pytest

from util import NCPU, R, circuit_arrays, fr_bytes, rand_fr_array

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ints(xs):
    return [int(v, 16) for v in xs]


def _case_inputs(c):
    wL, wR, wO = ([_ints(r) for r in c[k]] for k in ("wL", "wR", "wO"))
    return (wL, wR, wO, _ints(c["cs"])), (_ints(c["aL"]), _ints(c["aR"]), _ints(c["aO"])), _ints(c["transcript"])


def test_constants(ref):
    assert ref.selfcheck()


def test_python_oracle_reproduces_goldens(ref):
    cases = json.load(open(os.path.join(GOLD, "prove_small.json")))["cases"]
    for c in cases[:4]:          # the literal restatement is slow; the rest is covered through the C oracle
        circ, asg, tr = _case_inputs(c)
        srs = ref.SRS(c["d"], int(c["x"], 16), int(c["alpha"], 16))
        proof, _ = ref.prove(srs, asg, circ, tr)
        assert ref.proof_to_bytes(proof).hex() == c["proof"]


def test_c_oracle_matches_prove_goldens(orc):
    for c in json.load(open(os.path.join(GOLD, "prove_small.json")))["cases"]:
        circ, asg, tr = _case_inputs(c)
        srs = orc.SRS(c["d"], int(c["x"], 16), int(c["alpha"], 16), threads=NCPU)
        flat = lambda m: fr_bytes([v for r_ in m for v in r_])
        for mode in (0, 1):                   # reference-shaped fold, and Pippenger
            for use_ntt in (False, True):     # schoolbook (the reference's sparse convolution), and NTT
                orc.set_mode(mode, 4)
                got = orc.prove(srs, c["n"], c["Q"], flat(circ[0]), flat(circ[1]), flat(circ[2]), fr_bytes(circ[3]),
                                fr_bytes(asg[0]), fr_bytes(asg[1]), fr_bytes(asg[2]), fr_bytes(tr), use_ntt)
                assert got.hex() == c["proof"], (c["name"], mode, use_ntt)
        orc.set_mode(1, NCPU)


def test_c_oracle_matches_commitment_goldens(orc):
    g = json.load(open(os.path.join(GOLD, "commitment_small.json")))
    d = g["d"]
    srs = orc.SRS(d, int(g["x"], 16), int(g["alpha"], 16), threads=4)
    for k, v in g["srs"]["gNegativeX"].items():
        assert srs.points(0, -(int(k) + 1), 1)[0].tobytes().hex() == v          # SRS.hs:33
    for k, v in g["srs"]["gPositiveX"].items():
        assert srs.points(0, int(k), 1)[0].tobytes().hex() == v                  # SRS.hs:34
    for k, v in g["srs"]["gNegativeAlphaX"].items():
        assert srs.points(1, -(int(k) + 1), 1)[0].tobytes().hex() == v           # SRS.hs:37
    for k, v in g["srs"]["gPositiveAlphaX"].items():
        assert srs.points(1, int(k) + 1, 1)[0].tobytes().hex() == v              # SRS.hs:39
    assert srs.points(1, 0, 1)[0].tobytes() == bytes(96)                          # g^alpha omitted, SRS.hs:38
    for p in g["polys"]:
        exps = np.array([e for e, _ in p["terms"]], np.int64)
        co = fr_bytes([int(c, 16) for _, c in p["terms"]])
        assert orc.commit_poly(srs, p["max"], exps, co).hex() == p["commit"]
        fz, W = orc.open_poly(srs, int(p["z"], 16), exps, co)
        assert fz == int(p["fz"], 16) and W.hex() == p["open"]
    gen = orc.g1_gen()
    for m in g["gen_multiples"]:
        assert orc.g1_mul(gen, int(m["k"], 16)).hex() == m["point"]


def test_c_oracle_matches_python_on_fresh_cases(orc, ref):
    pyr = random.Random(1234)
    for n, Q in [(1, 1), (2, 1), (4, 4), (6, 2)]:
        circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
        d = max(7 * n, 4 * n + 8) + pyr.randrange(0, 4)
        x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
        tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
        proof, _ = ref.prove(ref.SRS(d, x, alpha), asg, circ, tr)
        got = orc.prove(orc.SRS(d, x, alpha, threads=4), n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(tr))
        assert got == ref.proof_to_bytes(proof)


def test_public_known_answer_vectors(orc, ref):
    """EIP-2537's doubles of the standard generators (tests/golden/eip2537_kat.json: public vectors, independent of this
    repository and of the reference, which holds none): both oracles and the pairing oracle's G2 reproduce them"""
    from oracle import pairing as pg
    k = json.load(open(os.path.join(GOLD, "eip2537_kat.json")))
    g = (int(k["g1_generator"]["x"], 16), int(k["g1_generator"]["y"], 16))
    g2x = (int(k["g1_generator_doubled"]["x"], 16), int(k["g1_generator_doubled"]["y"], 16))
    assert ref.G1_GEN == g and ref.g1_add(g, g) == g2x and ref.g1_mul(g, 2) == g2x
    enc = lambda p: p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little")
    assert orc.g1_mul(enc(g), 2) == enc(g2x)
    h2 = k["g2_generator_doubled"]
    want = ((int(h2["x_c0"], 16), int(h2["x_c1"], 16)), (int(h2["y_c0"], 16), int(h2["y_c1"], 16)))
    assert pg.g2_add(pg.G2_GEN, pg.G2_GEN) == want and pg.g2_mul(pg.G2_GEN, 2) == want


# ---- the reference's properties, restated ----------------------------------------------------
def test_prop_linear_constraints(ref):
    """test/Test/Constraints.hs:19-27"""
    pyr = random.Random(1)
    for _ in range(10):
        n = pyr.randint(1, 20)
        (wL, wR, wO, cs), (aL, aR, aO) = ref.rnd_circuit(pyr, n, pyr.randint(1, n))
        dot = lambda a, b: sum(x * y for x, y in zip(a, b)) % R
        assert all((dot(aL, wL[i]) + dot(aR, wR[i]) + dot(aO, wO[i])) % R == cs[i] for i in range(len(cs)))


def test_prop_rpoly(ref):
    """r(X,Y) at (x,y) == r(XY,1) at xy (test/Test/Constraints.hs:29-34): why scaling c_e by y^e is evalY"""
    pyr = random.Random(2)
    aL = [pyr.randrange(R) for _ in range(3)]; aR = [pyr.randrange(R) for _ in range(3)]
    rP = ref.r_poly(aL, aR, [a * b % R for a, b in zip(aL, aR)])
    x, y = pyr.randrange(1, R), pyr.randrange(1, R)
    assert ref.lp_eval(ref.eval_y(y, rP), x) == ref.lp_eval(ref.eval_y(1, rP), x * y % R)


def test_prop_zero_constant_terms(ref):
    """test/Test/Constraints.hs:37-83: no X^0 term in r, s, r+s; X^0 coefficient of t(X,Y) is zero"""
    pyr = random.Random(3)
    for _ in range(4):
        n = pyr.randint(1, 8)
        (wL, wR, wO, cs), (aL, aR, aO) = ref.rnd_circuit(pyr, n, pyr.randint(1, n))
        rXY, sXY = ref.r_poly(aL, aR, aO), ref.s_poly(wL, wR, wO)
        assert 0 not in rXY and 0 not in sXY and 0 not in ref.biv_add(rXY, sXY)
        assert 0 not in ref.t_poly(rXY, sXY, ref.k_poly(cs, n))


def test_prop_commitment_scheme(ref):
    """test/Test/CommitmentScheme.hs:25-96: pcV (commit, open) for t(X,y) with max = d and r(X,1) with
    max = n at z and at yz -- the pairing equation as its Fr identity on known discrete logs"""
    pyr = random.Random(4)
    for _ in range(3):
        n = pyr.randint(1, 5)
        (wL, wR, wO, cs), (aL, aR, aO) = ref.rnd_circuit(pyr, n, pyr.randint(1, n))
        d = {1: 12, 2: 16}.get(n, 7 * n) + pyr.randrange(3)
        x, y, z, alpha = (pyr.randrange(1, R) for _ in range(4))
        srs = ref.SRS(d, x, alpha)
        rXY = ref.r_poly(aL, aR, aO)
        for maxm, fX, zz in [(d, ref.eval_y(y, ref.t_poly(rXY, ref.s_poly(wL, wR, wO), ref.k_poly(cs, n))), z),
                             (n, ref.eval_y(1, rXY), z), (n, ref.eval_y(1, rXY), y * z % R)]:
            F = ref.commit_poly(srs, maxm, fX)
            v, W = ref.open_poly(srs, zz, fX)
            Fl, Wl = ref.commit_log(srs, maxm, fX), ref.open_log(srs, zz, fX)
            assert F == ref.g1_mul(ref.G1_GEN, Fl) and W == ref.g1_mul(ref.G1_GEN, Wl)
            assert ref.pcv_logs(srs, maxm, Fl, zz, v, Wl)
            assert not ref.pcv_logs(srs, maxm, Fl, zz, (v + 1) % R, Wl)


def test_prop_protocol_accepts(ref):
    """test/Test/Protocol.hs:14-23 and test/Test/Signature.hs:20-36: verify (prove ...) holds"""
    pyr = random.Random(5)
    for n, Q in [(1, 1), (3, 2)]:
        circ, asg = ref.rnd_circuit(pyr, n, Q)
        d = {1: 12, 2: 16}.get(n, 7 * n) + pyr.randrange(5)
        srs = ref.SRS(d, pyr.randrange(1, R), pyr.randrange(1, R))
        tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
        proof, _ = ref.prove(srs, asg, circ, tr)
        assert ref.verify_exponent(srs, circ, asg, tr, proof)
        bad = dict(proof); bad["prA"] = (proof["prA"] + 1) % R
        assert not ref.verify_exponent(srs, circ, asg, tr, bad)


def test_error_contract(orc, ref):
    """d < 7n panics (Protocol.hs:54-55); n = 1, 2 need d >= 12, 16 (test/Test/Reference.hs:97-103)
    because t(X,y) reaches X^{-4n-8}; an unsatisfied circuit puts a constant term into t -> index -1"""
    pyr = random.Random(6)
    circ, asg, enc = circuit_arrays(ref, pyr, 3, 2)
    tr = fr_bytes([pyr.randrange(1, R) for _ in range(12)])
    args = (3, 2, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], tr)
    with pytest.raises(orc.OracleError) as e:
        orc.prove(orc.SRS(20, 5, 7, threads=2), *args)
    assert e.value.code == 1
    with pytest.raises(ref.DTooSmall):
        ref.prove(ref.SRS(20, 5, 7), asg, circ, [1] * 12)
    c1, a1, e1 = circuit_arrays(ref, pyr, 1, 1)
    tr1 = fr_bytes([pyr.randrange(1, R) for _ in range(10)])
    with pytest.raises(orc.OracleError) as e:
        orc.prove(orc.SRS(11, 5, 7, threads=2), 1, 1, e1["wL"], e1["wR"], e1["wO"], e1["cs"], e1["aL"], e1["aR"], e1["aO"], tr1)
    assert e.value.code == 2
    with pytest.raises(IndexError):
        ref.prove(ref.SRS(11, 5, 7), a1, c1, [pyr.randrange(1, R) for _ in range(10)])
    orc.prove(orc.SRS(12, 5, 7, threads=2), 1, 1, e1["wL"], e1["wR"], e1["wO"], e1["cs"], e1["aL"], e1["aR"], e1["aO"], tr1)
    bad = enc["cs"].copy(); bad[0, 0] ^= 1
    with pytest.raises(orc.OracleError) as e:
        orc.prove(orc.SRS(30, 5, 7, threads=2), 3, 2, enc["wL"], enc["wR"], enc["wO"], bad, enc["aL"], enc["aR"], enc["aO"], tr)
    assert e.value.code == 2


def test_open_poly_at_zero_both_oracles(orc, ref):
    """openPoly at z = 0 (CommitmentScheme.hs:43-48): f(0) = c_0 and the quotient is (f - c_0)/X -- X is a unit of the
    Laurent ring, so `divide` is exact; the python restatement, the C oracle and a direct group computation agree"""
    d, x, alpha = 40, 1234567, 7654321
    s, o = ref.SRS(d, x, alpha), orc.SRS(d, x, alpha, threads=2)
    assert ref.open_poly(s, 0, {1: 5}) == (0, ref.g1_mul(ref.G1_GEN, 5))
    pyr = random.Random(17)
    for f in ({0: 7}, {0: 7, 1: 3}, {3: 9}, {2: 1, 5: R - 1, 30: 12345}, {e: pyr.randrange(1, R) for e in range(0, 35, 3)}):
        fz, W = ref.open_poly(s, 0, f)
        assert fz == f.get(0, 0)
        want = ref.INF
        for e, c in f.items():
            if e > 0:
                want = ref.g1_add(want, ref.g1_mul(s.gPositiveX(e - 1), c))
        assert W == want
        exps = np.array(sorted(f), np.int64)
        ofz, oW = orc.open_poly(o, 0, exps, fr_bytes([f[e] for e in sorted(f)]))
        assert ofz == fz and oW == (bytes(96) if W is None else W[0].to_bytes(48, "little") + W[1].to_bytes(48, "little"))
    with pytest.raises(orc.OracleError) as e:
        orc.open_poly(o, 0, np.array([-2, 1], np.int64), fr_bytes([5, 1]))
    assert e.value.code == 4


def test_oracle_msm_flavours_agree(orc):
    """reference-shaped fold == threaded Pippenger, incl. structured scalars"""
    srs = orc.SRS(600, 3, 5, threads=4)
    g = np.random.default_rng(0)
    sc = rand_fr_array(g, 1000)
    sc[100:400] = sc[100]            # many equal coefficients, as in s(X,y) with an all-ones row
    sc[500:520] = 0
    assert orc.msm_srs(srs, 0, -500, sc, 0, 1) == orc.msm_srs(srs, 0, -500, sc, 1, 7)
    assert orc.msm_srs(srs, 1, 1, sc[:600], 0, 1) == orc.msm_srs(srs, 1, 1, sc[:600], 1, 32)


def test_oracle_ntt_roundtrip_and_product(orc):
    g = np.random.default_rng(1)
    a = rand_fr_array(g, 256)
    assert np.array_equal(orc.ntt(orc.ntt(a), True), a)
    b = rand_fr_array(g, 100)
    assert np.array_equal(orc.poly_mul(a, b, True), orc.poly_mul(a, b, False))


# ---- the reference's acceptance tests with real pairings (oracle/pairing.py) ---------------------
def test_pairing_selfcheck():
    from oracle import pairing
    assert pairing.selfcheck()


def test_verify_accepts_prove_with_pairings(ref):
    """test/Test/Protocol.hs:14-23: verify srs circuit proof y z yzs == True for what prove produced, and
    test/Test/CommitmentScheme.hs:58-71: pcV on r(X,1) with max = n -- through e(.,.) on BLS12-381, not the trapdoor"""
    from oracle import pairing as pg
    pyr = random.Random(77)
    for circ, asg in (ref.arith_circuit_example1(), ref.rnd_circuit(pyr, 2, 1)):
        n, Q = len(asg[0]), len(circ[0])
        d = {1: 12, 2: 16}[n] + pyr.randrange(4)
        srs = pg.SRS(d, pyr.randrange(1, R), pyr.randrange(1, R))
        tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
        proof, ro = ref.prove(srs, asg, circ, tr)
        assert pg.verify(srs, circ, proof, ro["y"], ro["z"], ro["yzs"])
        bad = dict(proof); bad["prWt"] = ref.g1_add(proof["prWt"], ref.G1_GEN)
        assert not pg.verify(srs, circ, bad, ro["y"], ro["z"], ro["yzs"])
        rX1 = ref.eval_y(1, ref.r_poly(*asg))
        z = pyr.randrange(1, R)
        F, op = ref.commit_poly(srs, n, rX1), ref.open_poly(srs, z, rX1)
        assert pg.pc_v(srs, n, F, z, op)
        assert not pg.pc_v(srs, n, F, z, ((op[0] + 1) % R, op[1]))
