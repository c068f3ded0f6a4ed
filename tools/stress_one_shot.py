#!/usr/bin/env python3
"""Concurrency soak of the one-shot path: K independent statements (every one its own circuit, assignment and transcript) through
sonic_prove_many -- two host threads making sonic_prove calls into parked shells, the circuit uploaded inside the proof, the runs of equal
coefficients through the SRS's running sums -- must give, statement by statement, the bytes of a resident prepared handle proving them
one at a time; repeated with the runs switched off.
    python tools/stress_one_shot.py [--log2n 14] [--proofs 128] [--distinct 6]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sonic_amd  # noqa: E402
from sonic_amd import _lib  # noqa: E402
from util import big_circuit, rand_fr_array  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=14)
ap.add_argument("--proofs", type=int, default=128)
ap.add_argument("--distinct", type=int, default=6)
a = ap.parse_args()
_lib.check(_lib.lib().sonic_init(0))
n, Q = 1 << a.log2n, 2
rng = np.random.default_rng(1)
srs = sonic_amd.SRS.new(8 * n, 0xabcdef1234567, 0x7654321fedcba)
circs = [big_circuit(200 + i, n, Q, None) for i in range(a.distinct)]
trs = [rand_fr_array(rng, 8 + 2 * Q) for _ in range(a.proofs)]
for t in trs:
    t[:, 0] |= 1
want = []
for i, c in enumerate(circs):
    p = sonic_amd.Prover(srs, sonic_amd.ArithCircuit(sonic_amd.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"]))
    p.set_assignment(sonic_amd.Assignment(c["aL"], c["aR"], c["aO"]))
    want.append({k: p.prove_bytes(trs[k]) for k in range(i, a.proofs, a.distinct)})
    p.close()
sts = []
for k in range(a.proofs):
    c = circs[k % a.distinct]
    sts.append((sonic_amd.Assignment(c["aL"], c["aR"], c["aO"]), sonic_amd.ArithCircuit(sonic_amd.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"]), trs[k]))
for mode in ("1", "0"):
    os.environ["SONIC_PROVE_RUNS"] = mode
    t0 = time.perf_counter()
    out = sonic_amd.prove_many([srs], sts)
    dt = time.perf_counter() - t0
    bad = [k for k in range(a.proofs) if out[k] != want[k % a.distinct][k]]
    print(f"SONIC_PROVE_RUNS={mode}: {a.proofs} one-shot proofs of n=2^{a.log2n} over {a.distinct} circuits in {dt * 1e3:.0f} ms; mismatches: {bad}", flush=True)
    assert not bad
print("ok")
