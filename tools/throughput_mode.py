#!/usr/bin/env python3
"""Throughput mode (BASELINE.json configs[4]): a batch of independent proofs at n = 2^16 streamed through one GPU by
T host threads, each with its own prover handle (own streams / workspaces) on a shared SRS.
    python tools/throughput_mode.py [--log2n 16] [--proofs 16] [--threads 1 2 4]"""
import argparse, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sonic_amd
from sonic_amd import _lib
from util import big_circuit, rand_fr_array

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=16)
ap.add_argument("--proofs", type=int, default=16)
ap.add_argument("--threads", type=int, nargs="+", default=[1, 2, 4])
a = ap.parse_args()
_lib.check(_lib.lib().sonic_init(0))
n, Q = 1 << a.log2n, 2
rng = np.random.default_rng(0)
x = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
al = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
srs = sonic_amd.SRS.new(8 * n, x, al)
circ = big_circuit(1, n, Q, None)
trs = [rand_fr_array(rng, 8 + 2 * Q) for _ in range(a.proofs)]
for t in trs: t[:, 0] |= 1
for T in a.threads:
    provers = []
    for _ in range(T):
        p = sonic_amd.Prover(srs, sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"]))
        p.set_assignment(sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"]))
        p.prove_bytes(trs[0])
        provers.append(p)
    out = [None] * a.proofs
    def work(k):
        for i in range(k, a.proofs, T): out[i] = provers[k].prove_bytes(trs[i])
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"threads={T}: {a.proofs} proofs of n=2^{a.log2n} in {dt*1e3:.1f} ms -> {a.proofs/dt:.1f} proofs/s", flush=True)
    if T == a.threads[0]: ref = list(out)
    else: assert out == ref, "proofs differ between thread counts"
    for p in provers: p.close()
