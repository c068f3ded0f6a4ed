"""Opt-in Fiat-Shamir transcript on the GPU: sonic_prover_prove_fs derives every `rnd` draw of the reference
(src/Sonic/Protocol.hs:58,66,76,84-85; src/Sonic/Signature.hs:48,60) from the hash of what precedes it, in six passes over the
prover.  Byte parity against the python restatement (oracle/sonic_ref.py prove_fs, committed fixture), against the ordinary
one-pass prove on the transcript it reports, and acceptance / rejection through sonic_verify_fs (host pairings)."""
import json
import os
import random

import pytest

from util import NCPU, R, big_circuit, circuit_arrays, fr_bytes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FS = json.load(open(os.path.join(HERE, "golden", "fs_small.json")))["cases"]
BASE = {c["name"]: c for c in json.load(open(os.path.join(HERE, "golden", "prove_small.json")))["cases"]}


def _circuit(sonic, b):
    iv = lambda v: int(v, 16)    # noqa: E731
    m = lambda w: [[iv(v) for v in r] for r in w]    # noqa: E731
    circuit = sonic.ArithCircuit(sonic.GateWeights(m(b["wL"]), m(b["wR"]), m(b["wO"])), [iv(v) for v in b["cs"]])
    asg = sonic.Assignment([iv(v) for v in b["aL"]], [iv(v) for v in b["aR"]], [iv(v) for v in b["aO"]])
    return circuit, asg


@pytest.mark.parametrize("case", FS, ids=[c["name"] for c in FS])
def test_prove_fs_matches_fixture(sonic, case):
    b = BASE[case["name"]]
    srs = sonic.SRS.new(b["d"], int(b["x"], 16), int(b["alpha"], 16))
    circuit, asg = _circuit(sonic, b)
    assert sonic.fs_circuit_digest(circuit).hex() == case["circuit_digest"] and sonic.fs_srs_id(srs).hex() == case["srs_id"]
    for prepare in (False, True):
        p = sonic.Prover(srs, circuit, prepare=prepare)
        p.set_assignment(asg)
        raw, tr = p.prove_fs(bytes.fromhex(case["circuit_digest"]), bytes.fromhex(case["seed"]))
        assert raw.hex() == case["proof"] and ["%x" % v for v in tr] == case["transcript"]
        assert p.prove_bytes(tr) == raw                       # the one-pass prover on the reported transcript: same bytes
        p.close()
    proof = sonic.Proof.from_bytes(raw, b["Q"])
    assert sonic.verify_fs(srs, circuit, proof)
    o = sonic.fs_challenges(srs, circuit, proof)
    assert [o.rndOracleY, o.rndOracleZ] == tr[4:6] and sonic.verify(srs, circuit, proof, o.rndOracleY, o.rndOracleZ, o.rndOracleYZs)


def test_prove_fs_random_circuits_and_rejections(sonic, orc, ref):
    pyr = random.Random(99)
    for n, Q in ((1, 1), (5, 3), (64, 2)):
        d = 7 * n + pyr.randrange(0, 9)
        x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
        srs = sonic.SRS.new(d, x, alpha)
        circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
        circuit = sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3])
        seed = bytes(pyr.randrange(256) for _ in range(32))
        proof, oracle = sonic.prove_fs(srs, sonic.Assignment(*asg), circuit, seed)
        raw = proof.to_bytes()
        # the C oracle proves the same bytes from the transcript the hashes yield; python re-derives that transcript from the bytes
        rsrs = ref.SRS(d, x, alpha)
        assert sonic.fs_srs_id(srs) == ref.fs_srs_id(rsrs)
        y, z, ys, zs, u, v = ref.fs_challenges_of_proof(n, Q, d, ref.fs_circuit_digest(circ), ref.fs_srs_id(rsrs), raw)
        assert (oracle.rndOracleY, oracle.rndOracleZ, oracle.rndOracleYZs) == (y, z, list(zip(ys, zs)))
        assert (proof.prHscProof.hscU, proof.prHscProof.hscV) == (u, v)
        tr = ref.fs_blinders(seed, ref.fs_circuit_digest(circ), ref.fs_srs_id(rsrs), ref.fs_witness_digest(asg)) + [y, z] + ys + zs + [u, v]
        want = orc.prove(orc.SRS(d, x, alpha, threads=NCPU), n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(tr))
        assert raw == want
        assert sonic.verify_fs(srs, circuit, proof)
        # another seed: other blinders, another proof, still accepted
        proof2, _ = sonic.prove_fs(srs, sonic.Assignment(*asg), circuit, bytes(32))
        assert proof2.to_bytes() != raw and sonic.verify_fs(srs, circuit, proof2)
        # a proof that is not its own transcript's: rejected
        import copy
        bad = copy.deepcopy(proof)
        bad.prHscProof.hscV = (bad.prHscProof.hscV + 1) % R
        assert not sonic.verify_fs(srs, circuit, bad)
        bad = copy.deepcopy(proof)
        bad.prA = (bad.prA + 1) % R                       # changes every later challenge
        assert not sonic.verify_fs(srs, circuit, bad)
        # a statement that is not the proof's: rejected (the circuit digest opens the transcript)
        cs2 = list(circ[3])
        cs2[0] = (cs2[0] + 1) % R
        assert not sonic.verify_fs(srs, sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), cs2), proof)
        # another reference string of the same d: other challenges (the srs id opens the transcript), rejected
        srs2 = sonic.SRS.new(d, (x + 1) % R or 1, alpha)
        assert sonic.fs_challenges(srs2, circuit, proof).rndOracleY != oracle.rndOracleY and not sonic.verify_fs(srs2, circuit, proof)


def test_prove_fs_mid_size_and_error_contract(sonic):
    n, Q = 1 << 12, 2
    srs = sonic.SRS.new(8 * n, 0x1234567891, 0x987654321)
    circ = big_circuit(5, n, Q)
    circuit = sonic.ArithCircuit(sonic.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
    asg = sonic.Assignment(circ["aL"], circ["aR"], circ["aO"])
    digest = sonic.fs_circuit_digest(circuit)
    p = sonic.Prover(srs, circuit)
    p.set_assignment(asg)
    raw, tr = p.prove_fs(digest, b"\x07" * 32)
    assert p.prove_bytes(tr) == raw and p.prove_fs(digest, b"\x07" * 32)[0] == raw
    assert sonic.verify_fs(srs, circuit, sonic.Proof.from_bytes(raw, Q))
    # an assignment that does not satisfy the circuit: the constant term of t(X, y) is not zero -> the reference panics in
    # commitPoly's index (CommitmentScheme.hs:70-73); here the pass that commits T reports it
    aO = circ["aO"].copy()
    aO[0, 0] ^= 1
    p.set_assignment(sonic.Assignment(circ["aL"], circ["aR"], aO))
    with pytest.raises(sonic.SonicError) as e:
        p.prove_fs(digest, b"\x07" * 32)
    assert e.value.code == 2
    p.set_assignment(asg)
    assert p.prove_fs(digest, b"\x07" * 32)[0] == raw       # the handle is usable afterwards
    # one seed, two witnesses of the same circuit: the blinders must not repeat (with seed-only blinders R_1 - R_2 was an
    # unblinded commitment to the difference of the witnesses)
    aL2, aR2 = circ["aR"].copy(), circ["aL"].copy()          # aL and aR swapped: satisfies the same constraints when wL and wR carry the same row
    if circ["rows"][0] == circ["rows"][1]:
        p.set_assignment(sonic.Assignment(aL2, aR2, circ["aO"]))
        raw2, tr2 = p.prove_fs(digest, b"\x07" * 32)
        assert not set(tr2[:4]) & set(tr[:4]) and raw2 != raw
        assert sonic.verify_fs(srs, circuit, sonic.Proof.from_bytes(raw2, Q))
    with pytest.raises(ValueError):
        p.prove_fs(digest[:31], b"\x07" * 32)               # short inputs never reach the C side (it reads 32 bytes of each)
    with pytest.raises(ValueError):
        p.prove_fs(digest, b"\x07" * 5)
    p.close()
