#!/usr/bin/env python3
"""Time of verify (Protocol.hs:111-130: 4 + 3Q pcV checks on host threads, tower-field pairing of csrc/pairing.hpp) on proofs the
GPU made, at a few (n, Q).    python tools/verify_time.py"""
import time, sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import sonic_amd
from sonic_amd import _lib
from util import big_circuit, rand_fr_array
_lib.check(_lib.lib().sonic_init(0))
for lg, Q in ((10, 2), (14, 2), (12, 8)):
    n = 1 << lg
    rng = np.random.default_rng(1)
    x = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
    al = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
    srs = sonic_amd.SRS.new(8 * n, x, al)
    circ = big_circuit(1, n, Q, None)
    circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
    assignment = sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"])
    proof, orc = sonic_amd.prove(srs, assignment, circuit)
    ok = sonic_amd.verify(srs, circuit, proof, orc.rndOracleY, orc.rndOracleZ, orc.rndOracleYZs)
    t0 = time.perf_counter(); ok = sonic_amd.verify(srs, circuit, proof, orc.rndOracleY, orc.rndOracleZ, orc.rndOracleYZs); dt = time.perf_counter() - t0
    print(f"n=2^{lg} Q={Q}: verify -> {ok} in {dt*1e3:.1f} ms", flush=True)
