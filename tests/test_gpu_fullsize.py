"""Parity at BASELINE.json's full size (n = 2^18, d = 2^21) through a size-independent property: the
known-trapdoor exponent check.  The SRS is built from known x, alpha (as every reference test does,
test/Test/Protocol.hs:21), so every group element of a proof has a discrete log that is a polynomial
*evaluation*, computable in O(n) big-integer operations without any MSM:

    commitPoly srs max f = g^{alpha x^{d-max} f(x)}          openPoly srs z f = (f(z), g^{(f(x)-f(z))/(x-z)})

The GPU proof must equal, byte for byte, the proof rebuilt from those logs (one fixed-base scalar
multiplication per element on the CPU oracle) and the evaluations -- which is exactly what verify / pcV /
hscVerify accept (src/Sonic/Protocol.hs:111-130, src/Sonic/Signature.hs:74-90)."""
import os
import random

import numpy as np
import pytest

from util import NCPU, R, big_circuit, fr_bytes, rand_fr_array

pytestmark = pytest.mark.gpu


def geom(a, lo, hi):
    """sum_{i=lo..hi} a^i  (a != 1)"""
    return (pow(a, hi + 1, R) - pow(a, lo, R)) * pow(a - 1, -1, R) % R


class Evals:
    """evaluations of r'(X,Y), s(X,Y), k(Y) for the benchmark circuit (one all-ones row per weight matrix)"""

    def __init__(self, circ, n, Q, cns):
        self.n, self.Q, self.cns = n, Q, cns
        self.la, self.lb, self.lo = circ["ints"]
        self.rows, self.cs = circ["rows"], circ["cs_ints"]

    def r1(self, a):
        """r'(a, 1) = sum a_i a^i + b_i a^-i + c_i a^{-i-n} + sum c_{n+i} a^{-2n-i}"""
        n = self.n
        ai = pow(a, -1, R)
        acc, p, q = 0, 1, 1
        an = pow(ai, n, R)
        for i in range(n):
            p = p * a % R
            q = q * ai % R
            acc += self.la[i] * p + (self.lb[i] + self.lo[i] * an) * q
        acc %= R
        base = pow(ai, 2 * n, R)
        for i, c in enumerate(self.cns, start=1):
            acc += c * base * pow(ai, i, R)
        return acc % R

    def s(self, a, b):
        """s(a, b) (Constraints.hs:34-53) with wL, wR, wO = one all-ones row at rows[0..2]"""
        n = self.n
        ai, bi = pow(a, -1, R), pow(b, -1, R)
        rL, rR, rO = self.rows
        u = pow(b, n + rL + 1, R) * geom(ai, 1, n)
        v = pow(b, n + rR + 1, R) * geom(a, 1, n)
        an = pow(a, n, R)
        w = an * (pow(b, n + rO + 1, R) * geom(a, 1, n) - geom(a * b % R, 1, n) - geom(a * bi % R, 1, n))
        return (u + v + w) % R

    def k(self, b):
        return sum(c * pow(b, self.n + q + 1, R) for q, c in enumerate(self.cs)) % R


@pytest.mark.parametrize("log2n", [18, 20])      # BASELINE.json configs[2] (n = 2^18, d = 2^21) and configs[3] (n = 2^20, d = 2^23)
def test_prove_full_size_exponent_oracle(sonic, orc, log2n):
    n, Q = 1 << log2n, 2
    d = 8 * n
    pyr = random.Random(2026)
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    srs = sonic.SRS.new(d, x, alpha)
    circ = big_circuit(4242, n, Q, orc)
    tr = [pyr.randrange(2, R) for _ in range(8 + 2 * Q)]
    p = sonic.Prover(srs, sonic.ArithCircuit(sonic.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"]))
    p.set_assignment(sonic.Assignment(circ["aL"], circ["aR"], circ["aO"]))
    raw = p.prove_bytes(fr_bytes(tr))
    proof = sonic.Proof.from_bytes(raw, Q)
    if log2n == 18:
        # the same proof from a handle that did not commit the constraint rows: S_j then is one 3n-term MSM whose scalars are
        # 2n copies of two values (rndCircuit's all-ones rows) -- 26 buckets of 2^18 entries through the heavy-bucket path
        p2 = sonic.Prover(srs, sonic.ArithCircuit(sonic.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"]), prepare=False)
        p2.set_assignment(sonic.Assignment(circ["aL"], circ["aR"], circ["aO"]))
        assert p2.prove_bytes(fr_bytes(tr)) == raw
        p2.close()

    cns, y, z = tr[0:4], tr[4], tr[5]
    ys, zs, u, v = tr[6:6 + Q], tr[6 + Q:6 + 2 * Q], tr[6 + 2 * Q], tr[7 + 2 * Q]
    E = Evals(circ, n, Q, cns)
    g = orc.g1_gen()
    G = lambda k: sonic.g1_from_bytes(orc.g1_mul(g, k % R))
    inv = lambda a: pow(a, -1, R)
    xd = lambda maxm: pow(x, d - maxm, R)

    r_x, r_z, r_yz = E.r1(x), E.r1(z), E.r1(y * z % R)
    t_at = lambda a: (E.r1(a) * (E.r1(a * y % R) + E.s(a, y)) - E.k(y)) % R   # r(X,y) = r(Xy,1): test/Test/Constraints.hs:29-34
    t_x, t_z = t_at(x), t_at(z)
    assert proof.prR == G(alpha * xd(n) * r_x)                                  # Protocol.hs:63
    assert proof.prT == G(alpha * t_x)                                          # :73
    assert proof.prA == r_z and proof.prWa == G((r_x - r_z) * inv(x - z))       # :79
    assert proof.prB == r_yz and proof.prWb == G((r_x - r_yz) * inv(x - y * z)) # :80
    assert proof.prWt == G((t_x - t_z) * inv(x - z))                            # :81
    assert proof.prS == E.s(z, y)                                               # :83
    assert (proof.prA * (proof.prB + proof.prS) - E.k(y)) % R == t_z            # verify's t, Protocol.hs:120
    h = proof.prHscProof
    s_ux = E.s(u, x)
    assert h.hscC == G(alpha * s_ux)                                            # Signature.hs:52
    assert h.hscQv == G((s_ux - E.s(u, v)) * inv(x - v))                        # :63
    assert h.hscU == u and h.hscV == v
    for j in range(Q):
        s_xy, s_zy, s_uy = E.s(x, ys[j]), E.s(zs[j], ys[j]), E.s(u, ys[j])
        cm, (sj, wj) = h.hscS[j]
        sjp, wjp, qj = h.hscW[j]
        assert cm == G(alpha * s_xy)                                            # :42
        assert sj == s_zy and wj == G((s_xy - s_zy) * inv(x - zs[j]))           # :43
        assert sjp == s_uy and wjp == G((s_xy - s_uy) * inv(x - u))             # :54-55
        assert qj == G((s_ux - s_uy) * inv(x - ys[j]))                          # :55


@pytest.mark.parametrize("log2n", [14, 16])      # BASELINE.json configs[1] (n = 2^14) and configs[4] (n = 2^16)
def test_prove_bytes_vs_oracle_mid_sizes(sonic, orc, log2n):
    """d = 8n because the reference rejects d < 7n: the complete proof, byte for byte, against the C oracle -- the sizes at which
    the oracle itself (Pippenger + NTT on the host cores) still finishes in seconds"""
    n, Q = 1 << log2n, 2
    d = 8 * n
    pyr = random.Random(414)
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    g, o = sonic.SRS.new(d, x, alpha), orc.SRS(d, x, alpha, threads=NCPU)
    circ = big_circuit(77, n, Q, orc)
    tr = fr_bytes([pyr.randrange(2, R) for _ in range(8 + 2 * Q)])
    orc.set_mode(1, NCPU)
    want = orc.prove(o, n, Q, circ["wL"], circ["wR"], circ["wO"], circ["cs"], circ["aL"], circ["aR"], circ["aO"], tr, True)
    p = sonic.Prover(g, sonic.ArithCircuit(sonic.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"]))
    p.set_assignment(sonic.Assignment(circ["aL"], circ["aR"], circ["aO"]))
    assert p.prove_bytes(tr) == want
    # a slice of the SRS itself at this size, both bases
    for basis in (0, 1):
        assert np.array_equal(g.points(basis, d - 300, 300), o.points(basis, d - 300, 300))
        assert np.array_equal(g.points(basis, -d, 300), o.points(basis, -d, 300))


def test_prove_bytes_vs_oracle_at_the_bench_size(sonic, orc):
    """BASELINE.json configs[2] exactly as bench.py runs it (n = 2^18, Q = 2, d = 8n = 2^21): the complete proof, byte for byte, against
    the C oracle's prove() on the host cores (~20-40 s).  The oracle takes the SRS points from the GPU-made SRS (set-up is not part of
    prove(); generating 8 M points on the host would take minutes), so the GPU-made SRS is checked first: sampled elements against the
    oracle's own fixed-base multiples g^{x^e}, g^{alpha x^e}, then ALL of them through random linear combinations against the closed
    form in the exponent."""
    n, Q = 1 << 18, 2
    d = 8 * n
    pyr = random.Random(1818)
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    srs = sonic.SRS.new(d, x, alpha)
    g = orc.g1_gen()
    for e in [-d, -d + 1, -12345, -1, 0, 1, 2, 777777, d - 1, d] + [pyr.randrange(-d, d + 1) for _ in range(20)]:
        xe = pow(x, e, R)
        assert srs.points(0, e, 1)[0].tobytes() == orc.g1_mul(g, xe)
        if e != 0:
            assert srs.points(1, e, 1)[0].tobytes() == orc.g1_mul(g, alpha * xe % R)
    cores = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1), threads=cores)
    # ALL 2 (2d + 1) GPU-made points, not a sample: two random linear combinations per basis, summed by the ORACLE's Pippenger over
    # the bytes it was handed, against the closed form in the exponent.  rho_e = lambda^(e + d) for a random lambda: the two sides are
    # polynomials of degree 2d in lambda that agree at a random point only if every coefficient (every point) agrees (Schwartz-Zippel,
    # error 2d / r ~ 2^-233):  sum_e lambda^(e+d) g^{x^e} = g^{x^-d ((lambda x)^(2d+1) - 1) / (lambda x - 1)};  basis 1 has alpha and no e = 0.
    orc.set_mode(1, cores)
    gen = orc.g1_gen()
    for _ in range(2):
        lam = pyr.randrange(2, R)
        rho, v = [], 1
        for _i in range(2 * d + 1):
            rho.append(v)
            v = v * lam % R
        rho_b = fr_bytes(rho)
        total = pow(x, -d, R) * geom(lam * x % R, 0, 2 * d) % R
        assert orc.msm_srs(osrs, 0, -d, rho_b, 1, cores) == orc.g1_mul(gen, total)
        assert orc.msm_srs(osrs, 1, -d, rho_b, 1, cores) == orc.g1_mul(gen, alpha * (total - pow(lam, d, R)) % R)
    circ = big_circuit(1818, n, Q)
    tr = fr_bytes([pyr.randrange(2, R) for _ in range(8 + 2 * Q)])
    orc.set_mode(1, cores)
    want = orc.prove(osrs, n, Q, circ["wL"], circ["wR"], circ["wO"], circ["cs"], circ["aL"], circ["aR"], circ["aO"], tr, True)
    pr = sonic.Prover(srs, sonic.ArithCircuit(sonic.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"]))
    pr.set_assignment(sonic.Assignment(circ["aL"], circ["aR"], circ["aO"]))
    assert pr.prove_bytes(tr) == want
    pr.close()


def test_msm_full_size_properties(sonic, orc):
    """N = 2^20 MSM over an SRS slice: MSM(2a) == 2 MSM(a) (linearity), and the trapdoor value
    g^{alpha x sum_i (x0 x)^i} for the geometric scalar vector 1, x0, x0^2, ... in closed form"""
    from sonic_amd.commitment import msm_g1_srs
    d = 1 << 19
    pyr = random.Random(7)
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    srs = sonic.SRS.new(d, x, alpha)
    N = 1 << 20
    a = rand_fr_array(np.random.default_rng(9), N)
    A = msm_g1_srs(srs, 0, -d, a)
    a2 = fr_bytes([2 * int.from_bytes(a[i].tobytes(), "little") % R for i in range(N)])
    assert msm_g1_srs(srs, 0, -d, a2) == orc.g1_add(A, A)
    x0 = pyr.randrange(2, R)
    M = 1 << 16
    pw = [1] * M
    for i in range(1, M):
        pw[i] = pw[i - 1] * x0 % R
    got = msm_g1_srs(srs, 1, 1, fr_bytes(pw))
    assert got == orc.g1_mul(orc.g1_gen(), alpha * x % R * geom(x0 * x % R, 0, M - 1) % R)


@pytest.mark.parametrize("log2n", [20, 21, 24])
def test_ntt_full_size_closed_form(sonic, log2n):
    """the transforms at the bench size (2^21: ten wide stages in two passes of five), at 2^20 (nine: 5 + 4) and with three wide passes
    (2^24: 5 + 4 + 4 stages),
    beyond what the CPU oracle transforms in seconds: the transform of an input with three non-zero coefficients has the closed
    form out[i] = sum_k c_k w^(k i), w = 7^((r-1)/2^log2n) (include/sonic_hip.h); sampled positions against python integers, and
    the inverse transform of the result must be the input again.  (The product of src/Sonic/Constraints.hs:61 runs on these
    transforms; the assembly butterflies keep values in [0, 2r) between the stages.)"""
    from sonic_amd import _lib
    n = 1 << log2n
    pyr = random.Random(log2n)
    w = pow(7, (R - 1) >> log2n, R)
    terms = {0: R - 1, pyr.randrange(1, n): pyr.randrange(1, R), n - 1: pyr.randrange(1, R)}
    a = np.zeros((n, 32), np.uint8)
    for k, c in terms.items():
        a[k] = np.frombuffer(c.to_bytes(32, "little"), np.uint8)
    got = a.copy()
    _lib.check(_lib.lib().sonic_ntt_fr(got.ctypes.data, log2n, 0))
    for i in [0, 1, 2, n // 2, n - 1, 2047, 2048, 2049] + [pyr.randrange(n) for _ in range(56)]:
        want = sum(c * pow(w, k * i, R) for k, c in terms.items()) % R
        assert int.from_bytes(got[i].tobytes(), "little") == want, (log2n, i)
    _lib.check(_lib.lib().sonic_ntt_fr(got.ctypes.data, log2n, 1))
    assert np.array_equal(got, a)


def test_product_at_the_transform_limit_2p28(sonic):
    """the largest transform the butterfly routines address (2^28 points: the first stage's last twiddle sits at byte offset 2^32 - 32
    of its table; ADVICE r04 -- round 4 stopped at 2^27): the product of two sparse polynomials of 2^27 coefficients each, operands
    and result resident in HBM (sonic_poly_mul_fr_dev; 8 GB result, 16 GB of stage-major twiddles), has nine non-zero coefficients
    at the sums of the operands' positions -- checked there and at sampled positions that must be zero (the `*` of
    src/Sonic/Constraints.hs:61 at a size no CPU oracle reaches in a test)."""
    import ctypes as C
    from sonic_amd import _lib
    L = _lib.lib()
    na = nb = 1 << 27
    pyr = random.Random(28)
    ta = {0: pyr.randrange(1, R), pyr.randrange(1, na - 1): pyr.randrange(1, R), na - 1: R - 1}
    tb = {0: R - 2, pyr.randrange(1, nb - 1): pyr.randrange(1, R), nb - 1: pyr.randrange(1, R)}
    da, db, do = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for ptr, cnt in ((da, na), (db, nb), (do, na + nb - 1)):
        _lib.check(L.sonic_dev_alloc(32 * cnt, C.byref(ptr)))
    CH = 1 << 21                                         # 64-MB pieces of zeros; the few non-zero coefficients go on top
    zeros = np.zeros((CH, 32), np.uint8)
    for ptr, cnt, terms in ((da, na, ta), (db, nb, tb)):
        for o in range(0, cnt, CH):
            _lib.check(L.sonic_dev_upload(C.c_void_p(ptr.value + 32 * o), zeros.ctypes.data, 32 * min(CH, cnt - o)))
        for k, c in terms.items():
            _lib.check(L.sonic_dev_upload(C.c_void_p(ptr.value + 32 * k), c.to_bytes(32, "little"), 32))
    _lib.check(L.sonic_poly_mul_fr_dev(da, na, db, nb, do))
    want = {}
    for i, ci in ta.items():
        for j, cj in tb.items():
            want[i + j] = (want.get(i + j, 0) + ci * cj) % R
    one = C.create_string_buffer(32)

    def at(k):
        _lib.check(L.sonic_dev_download(one, C.c_void_p(do.value + 32 * k), 32))
        return int.from_bytes(one.raw, "little")
    for k, c in want.items():
        assert at(k) == c, k
    for k in [1, 2, 2047, 2048, na - 2, na, na + nb - 3] + [pyr.randrange(na + nb - 1) for _ in range(40)]:
        if k not in want:
            assert at(k) == 0, k
    for ptr in (da, db, do):
        L.sonic_dev_free(ptr)


def _dense_circuit(seed, n, Q):
    """dense random weights: every w_L/R/O[q][i] uniform in Fr (the reference's rndCircuit only ever has one all-ones row per matrix,
    test/Test/Reference.hs:141-155), c_q = w_L a_L + w_R a_R + w_O a_O (:138), a_O = a_L * a_R"""
    rng = np.random.default_rng(seed)
    c = big_circuit(seed, n, Q)
    la, lb, lo = c["ints"]
    W = [rand_fr_array(rng, Q * n) for _ in range(3)]
    cs = []
    for q in range(Q):
        acc = 0
        for w, a in zip(W, (la, lb, lo)):
            row = w[q * n:(q + 1) * n]
            acc += sum(int.from_bytes(row[i].tobytes(), "little") * a[i] for i in range(n))
        cs.append(acc % R)
    return dict(c, wL=W[0], wR=W[1], wO=W[2], cs=fr_bytes(cs))


@pytest.mark.parametrize("n", [3 * (1 << 13) + 5, (1 << 15) - 1])
def test_prove_bytes_off_the_power_of_two_grid(sonic, orc, n):
    """byte parity against the C oracle at sizes where the boundaries are crossed TOGETHER -- n = 3 * 2^13 + 5
    (7n + 9 = 172 108: a product of 2^18 points for 66 % padding; MSM sizes that are no multiple of any tile) and n = 2^15 - 1
    (7n + 9 = 229 378 just under 2^18) -- with Q = 4 DENSE random weights (every S_j MSM then has 3n distinct scalars, the prepared row
    tables C_q are real 29-window tables over 4 committed rows) prepared and not, and once with the rndCircuit rows (n equal
    coefficients in s(X,y): the heavy-bucket path) (src/Sonic/Protocol.hs:47-109, Signature.hs:38-72, Constraints.hs:34-68)."""
    pyr = random.Random(n)
    d = 7 * n + 3                                              # the smallest legal d is 7n (Protocol.hs:54); not a power of two either
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    srs = sonic.SRS.new(d, x, alpha)
    osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1))
    orc.set_mode(1, NCPU)
    for kind, Q in (("dense", 4), ("rnd", 2)):
        c = _dense_circuit(n, n, Q) if kind == "dense" else big_circuit(n + 1, n, Q)
        circuit = sonic.ArithCircuit(sonic.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"])
        asg = sonic.Assignment(c["aL"], c["aR"], c["aO"])
        tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
        want = orc.prove(osrs, n, Q, c["wL"], c["wR"], c["wO"], c["cs"], c["aL"], c["aR"], c["aO"], tr, True)
        for prepare in (False, True):
            p = sonic.Prover(srs, circuit, prepare=prepare)
            p.set_assignment(asg)
            assert p.prove_bytes(tr) == want, (n, kind, prepare)
            shares = []
            for r in range(3):                                 # and as three ranks' shares (cuts fall inside MSMs of odd lengths)
                p.set_share(r, 3)
                shares.append(p.prove_share(tr))
            assert sonic.proof_from_shares(Q, shares, tr) == want, (n, kind, prepare, "shares")
            p.close()
        proof, oracle_ = sonic.prove(srs, asg, circuit, transcript=[int.from_bytes(tr[i].tobytes(), "little") for i in range(8 + 2 * Q)])
        assert proof.to_bytes() == want, (n, kind, "one-shot")
    srs.close()


def test_msm_protocol_shaped_scalars_2p20(sonic, orc):
    """a stand-alone N = 2^20 MSM whose scalars look like the protocol's (src/Sonic/Constraints.hs:34-53 through
    commitPoly, CommitmentScheme.hs:25-29): long runs of 0, of 1, of r - 1 (= -1: folded onto the negated point), of ONE repeated random
    value (n equal coefficients: a heavy bucket in every window), small values, and uniform ones -- against the oracle's Pippenger."""
    from sonic_amd.commitment import msm_g1_srs
    N = 1 << 20
    d = 1 << 20
    pyr = random.Random(20)
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    srs = sonic.SRS.new(d, x, alpha)
    osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1))
    rng = np.random.default_rng(20)
    sc = rand_fr_array(rng, N)
    one = np.frombuffer((1).to_bytes(32, "little"), np.uint8)
    minus1 = np.frombuffer((R - 1).to_bytes(32, "little"), np.uint8)
    rep = np.frombuffer(pyr.randrange(1, R).to_bytes(32, "little"), np.uint8)
    half = np.frombuffer(((R - 1) // 2).to_bytes(32, "little"), np.uint8)
    sc[0:150000] = 0
    sc[150000:300000] = one
    sc[300000:450000] = minus1
    sc[450000:700000] = rep                                     # 250 000 copies of one value: 13 heavy buckets
    sc[700000:700100] = half                                    # the fold's boundary (r - 1)/2 and its neighbour
    sc[700100:700200] = np.frombuffer(((R + 1) // 2).to_bytes(32, "little"), np.uint8)
    small = rng.integers(0, 1 << 16, size=100000)
    sc[700200:800200] = 0
    sc[700200:800200, 0] = (small & 0xff).astype(np.uint8)
    sc[700200:800200, 1] = (small >> 8).astype(np.uint8)
    for basis, e0 in ((0, -d), (1, 1)):
        got = msm_g1_srs(srs, basis, e0, sc)
        assert got == orc.msm_srs(osrs, basis, e0, sc, 1, NCPU), (basis, e0)
    srs.close()


@pytest.mark.parametrize("log2d,values,copies", [(16, 40, 1000), (16, 3, 20000), (12, 5, 1500)])
def test_msm_many_heavy_buckets(sonic, orc, log2d, values, copies):
    """The heavy-bucket path itself (k_heavy_accum / k_heavy_finish): `values` repeated scalars with run lengths from `copies` down to a
    third of it, scattered among uniform ones -- at d = 2^16 with 40 values more heavy buckets (40 values x 16 windows = 640) than one
    pass of the kernel's record scan takes (512), stretches that cross several bucket borders, buckets cut by several stretches -- and
    the same with the runs ALSO equal to their neighbours' negatives (r - v: the same bucket, opposite sign).  Against the oracle."""
    from sonic_amd.commitment import msm_g1_srs
    d = 1 << log2d
    pyr = random.Random(1000 + values)
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    srs = sonic.SRS.new(d, x, alpha)
    osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1))
    N = 2 * d
    rng = np.random.default_rng(values)
    sc = rand_fr_array(rng, N)
    perm = rng.permutation(N)
    pos = 0
    for v in range(values):
        val = pyr.randrange(1, R)
        cnt = copies - (2 * copies // 3) * v // max(values - 1, 1)
        idx = perm[pos:pos + cnt]
        pos += cnt
        sc[idx] = np.frombuffer(val.to_bytes(32, "little"), np.uint8)
        if v % 2:
            sc[idx[: cnt // 4]] = np.frombuffer((R - val).to_bytes(32, "little"), np.uint8)
    assert pos <= N
    for basis, e0 in ((0, -d), (1, 1 - d)):
        got = msm_g1_srs(srs, basis, e0, sc)
        assert got == orc.msm_srs(osrs, basis, e0, sc, 1, NCPU), (basis, e0)
    srs.close()
