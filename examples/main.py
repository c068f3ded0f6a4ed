#!/usr/bin/env python3
"""examples/Main.hs of the reference, over the MI355X path: the 5-constraint / 2-gate circuit of the Bulletproofs paper
(arithCircuitExample, examples/Main.hs:38-63), SRS.new with d = 25 n (bench/Main.hs:18-19), prove, verify.

    python examples/main.py            ->  Success: True

`sonicProtocol` in the reference (examples/Main.hs:65-74 via test/Test/Reference.hs) draws x and alpha, builds the SRS,
proves and verifies; so does this, with the prover on the GPU and the verifier's pairings on the host."""
import os
import secrets
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sonic_amd as S  # noqa: E402

R = S.R_MODULUS


def arith_circuit_example(z: int):
    """arithCircuitExample x z (examples/Main.hs:38-63): 5 linear constraints, 2 multiplication gates"""
    wL = [[0, 0], [1, 0], [0, 1], [0, 0], [0, 0]]
    wR = [[0, 0], [0, 0], [0, 0], [1, 0], [0, 1]]
    wO = [[1, R - 1], [0, 0], [0, 0], [0, 0], [0, 0]]
    cs = [0, (4 - z) % R, (9 - z) % R, (9 - z) % R, (4 - z) % R]
    aL = [(4 - z) % R, (9 - z) % R]
    aR = [(9 - z) % R, (4 - z) % R]
    aO = [a * b % R for a, b in zip(aL, aR)]
    return S.ArithCircuit(S.GateWeights(wL, wR, wO), cs), S.Assignment(aL, aR, aO)


def sonic_protocol(circuit, assignment, x: int) -> bool:
    n = len(assignment.aL)
    alpha = secrets.randbelow(R - 1) + 1
    srs = S.SRS.new(25 * n, x, alpha)
    proof, oracle = S.prove(srs, assignment, circuit)
    return S.verify(srs, circuit, proof, oracle.rndOracleY, oracle.rndOracleZ, oracle.rndOracleYZs)


def sonic_protocol_fs(circuit, assignment, x: int) -> bool:
    """the same with the opt-in Fiat-Shamir transcript: the proof carries its own challenges, the verifier recomputes them"""
    n = len(assignment.aL)
    srs = S.SRS.new(25 * n, x, secrets.randbelow(R - 1) + 1)
    proof, _oracle = S.prove_fs(srs, assignment, circuit)
    return S.verify_fs(srs, circuit, proof)


def run_example() -> bool:
    x = secrets.randbelow(R - 1) + 1
    z = secrets.randbelow(R)
    circuit, assignment = arith_circuit_example(z)
    ok = sonic_protocol(circuit, assignment, x)
    print(f"Success: {ok}")
    if "--fs" in sys.argv:
        fs = sonic_protocol_fs(circuit, assignment, x)
        print(f"Success (Fiat-Shamir transcript): {fs}")
        ok = ok and fs
    return ok


if __name__ == "__main__":
    sys.exit(0 if run_example() else 1)
