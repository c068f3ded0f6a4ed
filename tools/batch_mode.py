#!/usr/bin/env python3
"""BASELINE configs[4] through both hosts of the throughput mode on one GPU: sonic_prove_batch (two host threads inside the library, one per
handle) and the Python pipeline (one host thread, submit / collect over two handles); 64 proofs at n = 2^16, assignment resident.
    python tools/batch_mode.py [--log2n 16] [--proofs 64] [--reps 3]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sonic_amd  # noqa: E402
from sonic_amd import _lib  # noqa: E402
from sonic_amd.workload import big_circuit, rand_fr_array  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=16)
ap.add_argument("--proofs", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
L = _lib.lib()
_lib.check(L.sonic_init(0))
n, Q = 1 << a.log2n, 2
rng = np.random.default_rng(0)
x = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
al = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
srs = sonic_amd.SRS.new(8 * n, x, al)
c = big_circuit(1, n, Q)
circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"])
asg = sonic_amd.Assignment(c["aL"], c["aR"], c["aO"])
trs = [rand_fr_array(rng, 8 + 2 * Q) for _ in range(a.proofs + 4)]
for t in trs:
    t[:, 0] |= 1
for rep in range(a.reps):
    hs = [sonic_amd.Prover(srs, circuit, prepare=True) for _ in range(2)]
    for h in hs:
        h.set_assignment(asg)
    sonic_amd.prove_batch(hs, trs[:4])
    L.sonic_device_sync()
    t0 = time.perf_counter()
    out = sonic_amd.prove_batch(hs, trs[4:])
    dt = time.perf_counter() - t0
    for h in hs:
        h.close()
    pipe = sonic_amd.ProverPipeline(srs, circuit, depth=2)
    pipe.set_assignment(asg)
    pipe.prove_all(trs[:4])
    L.sonic_device_sync()
    t0 = time.perf_counter()
    out2 = pipe.prove_all(trs[4:])
    dt2 = time.perf_counter() - t0
    pipe.close()
    assert out == out2
    print(f"n=2^{a.log2n}, {a.proofs} proofs: sonic_prove_batch (2 handles, 2 host threads) {a.proofs / dt:6.1f} proofs/s   pipeline (2 handles, 1 host thread) {a.proofs / dt2:6.1f} proofs/s", flush=True)
