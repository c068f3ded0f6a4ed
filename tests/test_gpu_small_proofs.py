"""Round 6: small proofs as ONE kernel chain (prove.hip `fused`), the batched openings, the four-stream handle, the in-proof assignment
upload of sonic_prove_batch / sonic_prove, sonic_one_shot_trim and the sampled "does the circuit have runs" hint -- every variant
must give the bytes of the CPU oracle (and therefore of each other).

Reference: Sonic.Protocol.prove (src/Sonic/Protocol.hs:47-109) with hscProve (src/Sonic/Signature.hs:38-72); the shapes follow
test/Test/Reference.hs:125-169 (rndCircuit) and bench/Main.hs:18-27 (d = 25 n, x = 1)."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from util import NCPU, R, big_circuit, circuit_arrays, fr_bytes, rand_fr_array

pytestmark = pytest.mark.gpu


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _circuit(sonic, circ):
    return sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3])


@pytest.mark.parametrize("n,Q", [(1, 1), (2, 5), (40, 2), (300, 3), (1000, 4), (5000, 2), (3 * 4096 + 5, 6)])
def test_one_chain_per_proof_matches_the_oracle_and_the_lanes(sonic, orc, ref, n, Q):
    """7 + 4Q = 11 .. 31 MSMs: one chunk, two chunks on two chain streams, prepared and not; against the C oracle and against the same
    handle configuration with the chain switched off (SONIC_PROVE_FUSED=0: round 5's one-chain-per-group lanes) and with six lanes of
    its own (SONIC_FUSED_LANES=6)"""
    pyr = random.Random(1000 * n + Q)
    d = max(7 * n, 12) + pyr.randrange(40)
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    g = sonic.SRS.new(d, x, alpha)
    o = orc.SRS(d, x, alpha, threads=NCPU)
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
    orc.set_mode(1, NCPU)
    want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], tr, n >= 256)
    for env in ({}, {"SONIC_PROVE_FUSED": 0}, {"SONIC_FUSED_LANES": 6}, {"SONIC_FUSED_LANES": 2}):
        with _env(**env):
            for prepare in (True, False):
                p = sonic.Prover(g, _circuit(sonic, circ), prepare=prepare)
                p.set_assignment(sonic.Assignment(*asg))
                for _ in range(2):                              # (the second proof runs in grown workspaces)
                    assert p.prove_bytes(tr) == want, (env, prepare)
                p.close()
    # the one-shot entry point (what sonic_amd.prove calls): host buffers per call, the shell parked between calls
    pr, _ = sonic.prove(g, sonic.Assignment(*asg), _circuit(sonic, circ), transcript=[int.from_bytes(tr[i].tobytes(), "little") for i in range(8 + 2 * Q)])
    assert pr.to_bytes() == want
    g.close()


def test_reference_bench_shape_x_equal_one(sonic, orc, ref):
    """bench/Main.hs:18-27: x = 1, alpha = 4, d = 25 n -- every SRS element of a basis is the same point, every bucket addition a
    doubling or a cancellation -- for the reference's two examples; proved through the chain over the window tables (an SRS this small
    ran without tables up to round 5)"""
    for (circ, asg) in (ref.arith_circuit_example1(), ref.arith_circuit_example2(12)):
        n, Q = len(asg[0]), len(circ[0])
        d = 25 * n
        g = sonic.SRS.new(d, 1, 4)
        o = orc.SRS(d, 1, 4, threads=NCPU)
        pyr = random.Random(n)
        tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
        enc = dict(wL=fr_bytes([v for r_ in circ[0] for v in r_]), wR=fr_bytes([v for r_ in circ[1] for v in r_]), wO=fr_bytes([v for r_ in circ[2] for v in r_]),
                   cs=fr_bytes(circ[3]), aL=fr_bytes(asg[0]), aR=fr_bytes(asg[1]), aO=fr_bytes(asg[2]))
        want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(tr))
        proof, ro = sonic.prove(g, sonic.Assignment(*asg), _circuit(sonic, circ), transcript=tr)
        assert proof.to_bytes() == want
        assert sonic.verify(g, _circuit(sonic, circ), proof, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
        # every MSM of an SRS with tables is planned over them, whatever its size (sonic_msm_plan: one shared bucket set)
        from sonic_amd import _lib
        pc, pw, pb = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().sonic_msm_plan(g._h, 3, C.byref(pc), C.byref(pw), C.byref(pb)))
        assert pb.value == 1 and pw.value * pc.value >= 255
        g.close()


def test_prove_batch_with_an_assignment_per_proof(sonic, orc):
    """sonic_prove_batch with K assignments handed over with the call (uploaded inside each proof's own queue since round 6): a circuit
    whose linear constraints are empty (all weights zero, cs = 0) is satisfied by ANY aL, aR with aO = aL o aR, so the K proofs really
    have K different witnesses; against set_assignment + prove on one handle and against the oracle.  A non-canonical element in ONE
    assignment fails that proof alone."""
    from sonic_amd import _lib
    n, Q, K = 3000, 2, 7
    d = 8 * n
    pyr = random.Random(77)
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    g = sonic.SRS.new(d, x, alpha)
    o = orc.SRS(d, x, alpha, threads=NCPU)
    zeros = np.zeros((Q * n, 32), np.uint8)
    cs = np.zeros((Q, 32), np.uint8)
    circuit = sonic.ArithCircuit(sonic.GateWeights(zeros, zeros, zeros), cs)
    rng = np.random.default_rng(5)
    asgs = []
    for k in range(K):
        aL, aR = rand_fr_array(rng, n), rand_fr_array(rng, n)
        la = [int.from_bytes(aL[i].tobytes(), "little") for i in range(n)]
        lb = [int.from_bytes(aR[i].tobytes(), "little") for i in range(n)]
        asgs.append(sonic.Assignment(aL, aR, fr_bytes([a * b % R for a, b in zip(la, lb)])))
    trs = [rand_fr_array(rng, 8 + 2 * Q) for _ in range(K)]
    for t in trs:
        t[:, 0] |= 1
    hs = [sonic.Prover(g, circuit, prepare=True) for _ in range(2)]
    got = sonic.prove_batch(hs, trs, assignments=asgs)
    assert len(set(got)) == K
    one = sonic.Prover(g, circuit, prepare=False)
    orc.set_mode(1, NCPU)
    for k in range(K):
        one.set_assignment(asgs[k])
        assert one.prove_bytes(trs[k]) == got[k], k
        if k in (0, K - 1):
            a = asgs[k]
            assert orc.prove(o, n, Q, zeros, zeros, zeros, cs, a.aL, a.aR, a.aO, trs[k], True) == got[k], k
    # a handle that proved with a per-call assignment keeps it (the next proof without one re-uses it)
    assert hs[(K - 1) % 2].prove_bytes(trs[K - 1]) == got[K - 1]
    # one bad witness: that proof reports SONIC_ERR_BAD_ENCODING (3), the others are made
    bad = [sonic.Assignment(a.aL.copy(), a.aR, a.aO) for a in asgs[:4]]
    bad[2].aL[5, :] = 0xff
    L = _lib.lib()
    tr = np.ascontiguousarray(np.stack(trs[:4]))
    aL = np.ascontiguousarray(np.stack([b.aL for b in bad])); aR = np.ascontiguousarray(np.stack([b.aR for b in bad])); aO = np.ascontiguousarray(np.stack([b.aO for b in bad]))
    out = np.zeros((4, L.sonic_proof_size(Q)), np.uint8)
    status = (C.c_int * 4)()
    arr = (C.c_void_p * 2)(*[h._h for h in hs])
    rc = L.sonic_prove_batch(arr, 2, 4, aL.ctypes.data, aR.ctypes.data, aO.ctypes.data, tr.ctypes.data, out.ctypes.data, status)
    assert rc == 3 and list(status) == [0, 0, 3, 0]
    assert [out[i].tobytes() for i in (0, 1, 3)] == [got[0], got[1], got[3]]
    # ... and the handle that met the bad witness refuses to prove on with it, until it is given a good one
    h_bad = hs[2 % 2]
    with pytest.raises(_lib.SonicError):
        h_bad.prove_bytes(trs[0])
    h_bad.set_assignment(asgs[0])
    assert h_bad.prove_bytes(trs[0]) == got[0]
    for h in hs + [one]:
        h.close()
    g.close()


def test_one_shot_trim_and_parked_shells(sonic, orc, ref):
    """sonic_prove parks the shell of a finished call; sonic_one_shot_trim frees the parked shells and the next call simply
    makes a new one"""
    from sonic_amd import _lib
    L = _lib.lib()
    L.sonic_one_shot_trim(-1)
    pyr = random.Random(9)
    n, Q = 64, 2
    d = 8 * n
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    g = sonic.SRS.new(d, x, alpha)
    o = orc.SRS(d, x, alpha, threads=NCPU)
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    for k in range(3):
        tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
        want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(tr))
        proof, _ = sonic.prove(g, sonic.Assignment(*asg), _circuit(sonic, circ), transcript=tr)
        assert proof.to_bytes() == want
        if k == 1:
            assert L.sonic_one_shot_trim(0) == 1              # the one shell these calls share
            assert L.sonic_one_shot_trim(-1) == 0
    g.close()                                                  # (frees the shell parked over it)
    assert L.sonic_one_shot_trim(-1) == 0


def test_runs_hint_follows_the_circuit(sonic, orc):
    """an unprepared handle takes runs of equal coefficients out of S_j only when a sample of the circuit's rows shows any (round 6): the
    reference's rndCircuit (all-ones rows) and a circuit of uniformly random weights both give the oracle's bytes, by whichever path"""
    n, Q = 1 << 16, 2
    d = 8 * n
    pyr = random.Random(66)
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    g = sonic.SRS.new(d, x, alpha)
    o = orc.SRS.from_points(d, g.points(0, -d, 2 * d + 1), g.points(1, -d, 2 * d + 1))
    orc.set_mode(1, NCPU)
    rng = np.random.default_rng(8)
    ones = big_circuit(61, n, Q)
    import bench
    dense = bench.dense_circuit(rand_fr_array, 62, n, Q)
    for c in (ones, dense):
        circuit = sonic.ArithCircuit(sonic.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"])
        tr = rand_fr_array(rng, 8 + 2 * Q)
        tr[:, 0] |= 1
        want = orc.prove(o, n, Q, c["wL"], c["wR"], c["wO"], c["cs"], c["aL"], c["aR"], c["aO"], tr, True)
        for prepare in (False, True):
            p = sonic.Prover(g, circuit, prepare=prepare)
            p.set_assignment(sonic.Assignment(c["aL"], c["aR"], c["aO"]))
            assert p.prove_bytes(tr) == want, prepare
            p.close()
    g.close()
