"""Randomised parity sweep of the MSM path (the fold of CommitmentScheme.hs:26-29,45-48): random lengths, slices, bases,
window plans and scalar patterns the protocol actually produces (zeros, +-1, long runs of one value, small values,
values near r), every result compared byte for byte with the CPU oracle."""
import random
import os
FUZZ_ROUNDS = int(os.environ.get("SONIC_FUZZ_ROUNDS", "1"))     # SONIC_FUZZ_ROUNDS=20 for a long sweep

import numpy as np
import pytest

from util import NCPU, R, fr_bytes, rand_fr_array

pytestmark = pytest.mark.gpu


def _scalars(pyr, n):
    kind = pyr.choice(["rand", "runs", "sparse", "small", "edge", "mixed"])
    if kind == "rand":
        return rand_fr_array(np.random.default_rng(pyr.randrange(1 << 30)), n)
    if kind == "runs":                      # s(X,y): blocks of one repeated coefficient
        vals, out = [pyr.randrange(R) for _ in range(pyr.randint(1, 4))], []
        while len(out) < n:
            out += [pyr.choice(vals)] * pyr.randint(1, max(1, n // 2))
        return fr_bytes(out[:n])
    if kind == "sparse":
        return fr_bytes([pyr.randrange(R) if pyr.random() < 0.05 else 0 for _ in range(n)])
    if kind == "small":
        return fr_bytes([pyr.randrange(0, 1 << pyr.choice([1, 8, 20, 40])) for _ in range(n)])
    if kind == "edge":
        pool = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, (R + 1) // 2, 1 << 253, (1 << 254) - 1, 1 << 64]
        return fr_bytes([pyr.choice(pool) for _ in range(n)])
    a = rand_fr_array(np.random.default_rng(pyr.randrange(1 << 30)), n)
    a[:: pyr.randint(2, 7)] = 0
    a[1:: pyr.randint(2, 9)] = np.frombuffer((R - 1).to_bytes(32, "little"), np.uint8)
    return a


@pytest.mark.parametrize("seed", range(6 * FUZZ_ROUNDS))
def test_msm_fuzz(sonic, orc, seed):
    from sonic_amd import _lib
    from sonic_amd.commitment import msm_g1_srs
    pyr = random.Random(1000 + seed)
    d = pyr.choice([300, 1 << 10, 5000, 1 << 13])
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    g, o = sonic.SRS.new(d, x, alpha), orc.SRS(d, x, alpha, threads=NCPU)
    try:
        for _ in range(12):
            n = pyr.choice([pyr.randint(0, 70), pyr.randint(71, 2 * d), 2 * d + 1, pyr.randint(1, 2 * d + 1)])
            e0 = pyr.randint(-d, d - n + 1) if n else 0
            basis = pyr.randint(0, 1)
            _lib.lib().sonic_msm_set_window(pyr.choice([0, 0, 0, 4, 5, 9, 13, 16]))
            sc = _scalars(pyr, n)
            assert msm_g1_srs(g, basis, e0, sc) == orc.msm_srs(o, basis, e0, sc, 1, NCPU), (seed, d, n, e0, basis)
    finally:
        _lib.lib().sonic_msm_set_window(0)


@pytest.mark.parametrize("seed", range(3 * FUZZ_ROUNDS))
def test_prove_fuzz(sonic, orc, ref, seed):
    """random (n, Q, d) in the range of the reference's own generator (rndCircuit: n <= 20, Q <= n; randomD) and beyond"""
    from util import circuit_arrays
    pyr = random.Random(2000 + seed)
    for _ in range(4):
        n = pyr.choice([pyr.randint(1, 20), pyr.randint(21, 300)])
        Q = pyr.randint(1, min(n, 6))
        d = {1: 12, 2: 16}.get(n, 7 * n) + pyr.randint(0, 3 * n)
        x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
        g, o = sonic.SRS.new(d, x, alpha), orc.SRS(d, x, alpha, threads=NCPU)
        circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
        tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
        want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], tr)
        p = sonic.Prover(g, sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3]))
        p.set_assignment(sonic.Assignment(*asg))
        assert p.prove_bytes(tr) == want, (seed, n, Q, d)
        assert p.prove_bytes(tr) == want          # deterministic on repeat


@pytest.mark.parametrize("seed", range(2 * FUZZ_ROUNDS))
def test_verify_fuzz(sonic, ref, seed):
    """verify . prove == True on random circuits (test/Test/Protocol.hs:14-23), and every single element of the proof matters:
    each of its 7 + 4Q points replaced by another point of the proof (a valid element of the subgroup) and each of its 5 + 2Q
    scalars moved by one is rejected -- the 4 + 3Q pairing checks and the scalar identities of verify / hscVerify
    (Protocol.hs:111-130, Signature.hs:74-90) through the tower-field pairing of csrc/pairing.hpp"""
    pyr = random.Random(5000 + seed)
    n = pyr.randint(1, 24)
    Q = pyr.randint(1, min(n, 3))
    d = {1: 12, 2: 16}.get(n, 7 * n) + pyr.randint(0, n)
    circ, asg = ref.rnd_circuit(pyr, n, Q)
    g = sonic.SRS.new(d, pyr.randrange(1, R), pyr.randrange(1, R))
    circuit = sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), circ[3])
    proof, ro = sonic.prove(g, sonic.Assignment(*asg), circuit, rng=pyr)
    args = (ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
    assert sonic.verify(g, circuit, proof, *args)
    raw = proof.to_bytes()
    kinds = list("GGFGFGGF") + list("GFG") * Q + list("FGG") * Q + list("GGFF")
    offs, o = [], 0
    for k in kinds:
        offs.append(o)
        o += 96 if k == "G" else 32
    assert o == len(raw)
    gslots = [i for i, k in enumerate(kinds) if k == "G"]
    for i, k in enumerate(kinds):
        b = bytearray(raw)
        if k == "G":
            j = gslots[(gslots.index(i) + 1) % len(gslots)]
            if raw[offs[j]:offs[j] + 96] == raw[offs[i]:offs[i] + 96]:
                continue
            b[offs[i]:offs[i] + 96] = raw[offs[j]:offs[j] + 96]
        else:
            v = (int.from_bytes(raw[offs[i]:offs[i] + 32], "little") + 1) % R
            b[offs[i]:offs[i] + 32] = v.to_bytes(32, "little")
        try:
            ok = sonic.verify(g, circuit, sonic.Proof.from_bytes(bytes(b), Q), *args)
        except sonic.SonicError:
            ok = False                                  # e.g. u or v moved onto 0
        assert not ok, (seed, n, Q, i, k)


def _random_d(pyr, n):
    """randomD (test/Test/Reference.hs:101-104)"""
    return pyr.randint(12, 100) if n == 1 else pyr.randint(16, 200) if n == 2 else pyr.randint(7 * n, 100 * n)


@pytest.mark.parametrize("seed", range(5 * FUZZ_ROUNDS))
def test_commitment_scheme_properties(sonic, ref, seed):
    """The three properties of test/Test/CommitmentScheme.hs on the reference's own generators (rndCircuit n <= 20, randomD,
    randomParams), with commitPoly / openPoly on the GPU and pcV through the product's pairing:
      :25-53  pcV srs d (commitPoly srs d t(X,y)) z (openPoly srs z t(X,y))
      :58-71  pcV srs n (commitPoly srs n r(X,1)) z (openPoly srs z r(X,1))
      :76-96  the same for the blinded r(X,1) + sum c_{n+i} X^{-2n-i}, opened at yz
    each also rejected for a moved evaluation.  (5 x SONIC_FUZZ_ROUNDS cases; the reference runs 25 / 50 / 50.)"""
    pyr = random.Random(9000 + seed)
    n = pyr.randint(1, 20)
    Q = pyr.randint(1, n)
    d = _random_d(pyr, n)
    x, y, z, alpha = (pyr.randrange(1, R) for _ in range(4))
    circ, asg = ref.rnd_circuit(pyr, n, Q)
    g = sonic.SRS.new(d, x, alpha)
    rXY = ref.r_poly(*asg)
    tXy = ref.eval_y(y, ref.t_poly(rXY, ref.s_poly(*circ[:3]), ref.k_poly(circ[3], n)))
    rX1 = ref.eval_y(1, rXY)
    blinded = ref.lp_add(rX1, {-2 * n - i: pyr.randrange(1, R) for i in range(1, 5)})
    for maxm, poly, pt in ((d, tXy, z), (n, rX1, z), (n, blinded, y * z % R)):
        F = sonic.commit_poly(g, maxm, poly)
        op = sonic.open_poly(g, pt, poly)
        assert op[0] == ref.lp_eval(poly, pt)
        assert sonic.pc_v(g, maxm, F, pt, op), (seed, n, Q, d, maxm)
        assert not sonic.pc_v(g, maxm, F, pt, ((op[0] + 1) % R, op[1]))
        assert not sonic.pc_v(g, maxm, F, (pt + 1) % R, op)


@pytest.mark.parametrize("seed", range(4 * FUZZ_ROUNDS))
def test_reference_acceptance_properties(sonic, ref, seed):
    """test/Test/Signature.hs:20-36 (hscVerify . hscProve, 20 cases in the reference) and test/Test/Protocol.hs:14-23
    (verify . prove, 50 cases) on the reference's generators -- rndCircuit (n <= 20, m <= n), randomD, randomParams -- with the
    prover on the GPU and the verifier's pairings on the host.  4 x SONIC_FUZZ_ROUNDS cases of each."""
    pyr = random.Random(12000 + seed)
    n = pyr.randint(1, 20)
    m = pyr.randint(1, n)
    d = _random_d(pyr, n)
    circ, asg = ref.rnd_circuit(pyr, n, m)
    g = sonic.SRS.new(d, pyr.randrange(1, R), pyr.randrange(1, R))
    circuit = sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), circ[3])
    yzs = [(pyr.randrange(1, R), pyr.randrange(1, R)) for _ in range(m)]
    hp = sonic.hsc_prove(g, circuit, yzs, rng=pyr)
    assert sonic.hsc_verify(g, circuit, yzs, hp), (seed, n, m, d)
    proof, ro = sonic.prove(g, sonic.Assignment(*asg), circuit, rng=pyr)
    assert sonic.verify(g, circuit, proof, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs), (seed, n, m, d)
