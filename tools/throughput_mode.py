#!/usr/bin/env python3
"""Throughput mode (BASELINE.json configs[4]): a batch of independent proofs at n = 2^16 streamed through one GPU by ONE host
thread over `depth` prover handles used in turn (sonic_prover_submit / sonic_prover_collect, sonic_amd.ProverPipeline) on a
shared SRS.  depth = 1 is one proof after the other.  Every depth must give the same bytes.
    python tools/throughput_mode.py [--log2n 16] [--proofs 64] [--depths 1 2 3]"""
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
ap.add_argument("--log2n", type=int, default=16)
ap.add_argument("--proofs", type=int, default=64)
ap.add_argument("--depths", type=int, nargs="+", default=[1, 2, 3])
ap.add_argument("--distinct", type=int, default=8, help="distinct circuits of the independent-statements leg (0: skip it)")
a = ap.parse_args()
_lib.check(_lib.lib().sonic_init(0))
n, Q = 1 << a.log2n, 2
rng = np.random.default_rng(0)
x = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
al = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
srs = sonic_amd.SRS.new(8 * n, x, al)
circ = big_circuit(1, n, Q, None)
circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
trs = [rand_fr_array(rng, 8 + 2 * Q) for _ in range(a.proofs)]
for t in trs:
    t[:, 0] |= 1
ref = None
for depth in a.depths:
    pipe = sonic_amd.ProverPipeline(srs, circuit, depth=depth)
    pipe.set_assignment(sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"]))
    pipe.prove_all(trs[:depth])                      # warm-up: grows every handle's workspaces
    _lib.lib().sonic_device_sync()
    t0 = time.perf_counter()
    out = pipe.prove_all(trs)
    dt = time.perf_counter() - t0
    print(f"depth={depth}: {a.proofs} proofs of n=2^{a.log2n} in {dt * 1e3:.1f} ms -> {a.proofs / dt:.1f} proofs/s", flush=True)
    if ref is None:
        ref = out
    else:
        assert out == ref, "proofs differ between pipeline depths"
    pipe.close()

# The batch read literally -- INDEPENDENT statements: every proof its own circuit, assignment and transcript, handed over as host
# buffers per proof (the reference's prove srs assignment circuit mapped over a list) through sonic_prove_many: two host threads on
# this GPU making one-shot calls into parked shells.  `--distinct` circuits are generated (python integers: ~0.2 s each at n = 2^16)
# and cycled through the batch; every proof still uploads its own copy.
if a.distinct > 0:
    circs = [big_circuit(100 + i, n, Q, None) for i in range(a.distinct)]
    sts = []
    for i in range(a.proofs):
        c = circs[i % a.distinct]
        sts.append((sonic_amd.Assignment(c["aL"], c["aR"], c["aO"]), sonic_amd.ArithCircuit(sonic_amd.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"]), trs[i]))
    sonic_amd.prove_many([srs], sts[:4])
    _lib.lib().sonic_device_sync()
    t0 = time.perf_counter()
    out = sonic_amd.prove_many([srs], sts)
    dt = time.perf_counter() - t0
    mb = (3 * Q * n + Q + 3 * n + 8 + 2 * Q) * 32 / 1e6
    print(f"independent statements (sonic_prove_many, {a.distinct} distinct circuits cycled, {mb:.0f} MB of host buffers per proof, unprepared): "
          f"{a.proofs} proofs of n=2^{a.log2n} in {dt * 1e3:.1f} ms -> {a.proofs / dt:.1f} proofs/s", flush=True)
    assert len(set(out)) == a.proofs
