#!/usr/bin/env python3
"""Generates tests/golden/*.json from oracle/sonic_ref.py (the literal big-integer restatement).

    python oracle/gen_golden.py

The reference holds no golden vectors and cannot be run here (see the header of sonic_ref.py), so
these fixtures pin the *restatement*: the C oracle and the HIP path are both checked against them.
Every fixture is data: inputs (SRS trapdoor, circuit, assignment, transcript) and expected outputs.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import sonic_ref as ref  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def hx(v):
    return "%x" % v


def pt(p):
    return ref.g1_to_bytes(p).hex()


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = random.Random(20261001)
    cases = []
    specs = [("example1", ref.arith_circuit_example1(), 25),          # bench/Main.hs: d = 25 n
             ("example2", ref.arith_circuit_example2(12), 50),
             ("rnd_n1", ref.rnd_circuit(rng, 1, 1), 12),                # test/Test/Reference.hs:102: n=1 needs d >= 12
             ("rnd_n2", ref.rnd_circuit(rng, 2, 2), 16),
             ("rnd_n3", ref.rnd_circuit(rng, 3, 2), 21),
             ("rnd_n8", ref.rnd_circuit(rng, 8, 3), 61)]
    for name, (circ, asg), d in specs:
        wL, wR, wO, cs = circ
        aL, aR, aO = asg
        Q = len(wL)
        x, alpha = rng.randrange(1, ref.R), rng.randrange(1, ref.R)
        tr = [rng.randrange(1, ref.R) for _ in range(8 + 2 * Q)]
        srs = ref.SRS(d, x, alpha)
        proof, _ = ref.prove(srs, asg, circ, tr)
        assert ref.verify_exponent(srs, circ, asg, tr, proof)
        cases.append({
            "name": name, "d": d, "x": hx(x), "alpha": hx(alpha), "n": len(aL), "Q": Q,
            "wL": [[hx(v) for v in r] for r in wL], "wR": [[hx(v) for v in r] for r in wR], "wO": [[hx(v) for v in r] for r in wO],
            "cs": [hx(v) for v in cs], "aL": [hx(v) for v in aL], "aR": [hx(v) for v in aR], "aO": [hx(v) for v in aO],
            "transcript": [hx(v) for v in tr], "proof": ref.proof_to_bytes(proof).hex(),
        })
    json.dump({"generator": "oracle/gen_golden.py", "cases": cases}, open(os.path.join(OUT, "prove_small.json"), "w"), indent=1)

    # SRS elements, commitPoly / openPoly on hand-made sparse Laurent polynomials, scalar multiples of g
    d = 40
    x, alpha = rng.randrange(1, ref.R), rng.randrange(1, ref.R)
    srs = ref.SRS(d, x, alpha)
    elems = {"gNegativeX": {k: pt(srs.gNegativeX(k)) for k in (0, 1, 17, d - 1)},
             "gPositiveX": {k: pt(srs.gPositiveX(k)) for k in (0, 1, 17, d)},
             "gNegativeAlphaX": {k: pt(srs.gNegativeAlphaX(k)) for k in (0, 1, 17, d - 1)},
             "gPositiveAlphaX": {k: pt(srs.gPositiveAlphaX(k)) for k in (0, 1, 17, d - 1)}}
    polys = []
    for lo, hi, maxm in [(-7, 5, d), (-30, 8, 10), (1, 6, d), (-6, -1, d), (-1, 1, d)]:
        hole = maxm - d
        f = {e: rng.randrange(1, ref.R) for e in range(lo, hi + 1) if e != hole and rng.random() < 0.8}
        z = rng.randrange(1, ref.R)
        fz, W = ref.open_poly(srs, z, f)
        polys.append({"max": maxm, "terms": [[e, hx(c)] for e, c in sorted(f.items())], "commit": pt(ref.commit_poly(srs, maxm, f)),
                      "z": hx(z), "fz": hx(fz), "open": pt(W)})
    muls = [{"k": hx(k), "point": pt(ref.g1_mul(ref.G1_GEN, k))} for k in [1, 2, 3, ref.R - 1, (ref.R - 1) // 2] + [rng.randrange(ref.R) for _ in range(4)]]
    json.dump({"generator": "oracle/gen_golden.py", "d": d, "x": hx(x), "alpha": hx(alpha), "srs": elems, "polys": polys, "gen_multiples": muls},
              open(os.path.join(OUT, "commitment_small.json"), "w"), indent=1)
    gen_fs()
    print("wrote", OUT)


def gen_fs():
    """Fiat-Shamir proofs (ref.prove_fs) of three of the circuits above: seed -> the transcript the hashes yield -> proof bytes"""
    base = json.load(open(os.path.join(OUT, "prove_small.json")))["cases"]
    cases = []
    for c in base:
        if c["name"] not in ("example1", "rnd_n2", "rnd_n8"):
            continue
        iv = lambda v: int(v, 16)    # noqa: E731
        circ = ([[iv(v) for v in r] for r in c["wL"]], [[iv(v) for v in r] for r in c["wR"]], [[iv(v) for v in r] for r in c["wO"]], [iv(v) for v in c["cs"]])
        asg = ([iv(v) for v in c["aL"]], [iv(v) for v in c["aR"]], [iv(v) for v in c["aO"]])
        srs = ref.SRS(c["d"], iv(c["x"]), iv(c["alpha"]))
        seed = bytes([len(c["name"])] * 32)
        proof, _o, tr = ref.prove_fs(srs, asg, circ, seed)
        assert ref.verify_exponent(srs, circ, asg, tr, proof)
        cases.append({"name": c["name"], "seed": seed.hex(), "circuit_digest": ref.fs_circuit_digest(circ).hex(),
                      "srs_id": ref.fs_srs_id(srs).hex(), "witness_digest": ref.fs_witness_digest(asg).hex(),
                      "transcript": [hx(v) for v in tr], "proof": ref.proof_to_bytes(proof).hex()})
    json.dump({"generator": "oracle/gen_golden.py (gen_fs): inputs are the cases of the same name in prove_small.json", "cases": cases},
              open(os.path.join(OUT, "fs_small.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
