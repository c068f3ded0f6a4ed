#!/usr/bin/env python3
"""The reference's own criterion benchmark shape (bench/Main.hs:18-50) as a timing: Example1 (n = 1, Q = 2) and Example2
(n = 2, Q = 2 ... as test/Test/Reference.hs defines them), x = 1, alpha = 4, d = 25 n; "Prover" = SRS.new + prove (the SRS is inside
the reference's timed closure, bench/Main.hs:45-47), "Verifier" = SRS.new + verify.  These are plumbing sizes: what they show is
the fixed cost of a call (launches, host tails, pairings), not throughput.  The circuits come from the oracle's restatement of the
examples (test infrastructure), the product does the proving and verifying.

  python tools/criterion_shape.py [--reps 20]
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import sonic_amd
    from sonic_amd import _lib
    from oracle import sonic_ref as ref
    R = ref.R
    _lib.check(_lib.lib().sonic_init(0))
    pyr = random.Random(2024)
    print("# reference criterion shape (bench/Main.hs): x = 1, alpha = 4, d = 25 n; ms per call on one MI355X, mean of", args.reps, "calls after 3 warm-ups")
    for name, (circ, asg) in (("Example1", ref.arith_circuit_example1()), ("Example2", ref.arith_circuit_example2(12))):
        n, Q = len(asg[0]), len(circ[0])
        d = 25 * n
        circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(*circ[:3]), circ[3])
        assignment = sonic_amd.Assignment(*asg)
        tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]

        def prover():
            srs = sonic_amd.SRS.new(d, 1, 4)
            return srs, sonic_amd.prove(srs, assignment, circuit, transcript=tr)

        def prove_only(srs):
            return sonic_amd.prove(srs, assignment, circuit, transcript=tr)

        for _ in range(3):
            srs, (proof, ro) = prover()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            srs, (proof, ro) = prover()
        t_prover = 1e3 * (time.perf_counter() - t0) / args.reps
        t0 = time.perf_counter()
        for _ in range(args.reps):
            prove_only(srs)
        t_prove = 1e3 * (time.perf_counter() - t0) / args.reps

        def verifier():
            s2 = sonic_amd.SRS.new(d, 1, 4)
            return sonic_amd.verify(s2, circuit, proof, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
        for _ in range(2):
            ok = verifier()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ok = verifier() and ok
        t_ver = 1e3 * (time.perf_counter() - t0) / args.reps
        t0 = time.perf_counter()
        for _ in range(args.reps):
            sonic_amd.verify(srs, circuit, proof, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
        t_ver_only = 1e3 * (time.perf_counter() - t0) / args.reps
        print(f"{name}: n = {n}, Q = {Q}, d = {d}:  Prover (SRS.new + prove) {t_prover:.2f} ms, prove alone {t_prove:.2f} ms;  "
              f"Verifier (SRS.new + verify, first use generates the G2 half) {t_ver:.2f} ms, verify alone {t_ver_only:.2f} ms;  accepted: {ok}")


if __name__ == "__main__":
    main()
