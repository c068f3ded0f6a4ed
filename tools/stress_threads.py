#!/usr/bin/env python3
"""Concurrency soak on one GPU: three host threads, each streaming 200 proofs (n = 2^14) through its own two-handle pipeline over
ONE shared SRS, must produce identical bytes; four threads, each 100 MSMs on its own lane, identical sums.
    python tools/stress_threads.py"""
import os, sys, threading, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import sonic_amd
from sonic_amd import _lib
from util import big_circuit, rand_fr_array
_lib.check(_lib.lib().sonic_init(0))
n, Q = 1 << 14, 2
rng = np.random.default_rng(0)
x = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
al = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
srs = sonic_amd.SRS.new(8 * n, x, al)
circ = big_circuit(1, n, Q, None)
circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
trs = [rand_fr_array(rng, 8 + 2 * Q) for _ in range(200)]
for t in trs: t[:, 0] |= 1
ref = None
res = {}
def work(k):
    pipe = sonic_amd.ProverPipeline(srs, circuit, depth=2)
    pipe.set_assignment(sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"]))
    res[k] = pipe.prove_all(trs)
    pipe.close()
t0 = time.time()
th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
[t.start() for t in th]; [t.join() for t in th]
print("3 threads x 200 proofs:", time.time() - t0, "s; identical:", res[0] == res[1] == res[2])
# MSM lanes from two threads
lanes_ok = []
sc = rand_fr_array(rng, 50000)
import ctypes as C
dp = C.c_void_p(); L = _lib.lib()
_lib.check(L.sonic_dev_alloc(32 * 50000, C.byref(dp))); _lib.check(L.sonic_dev_upload(dp, sc.ctypes.data, 32 * 50000))
def mwork(k):
    ln = sonic_amd.MsmLane(); out = []
    for i in range(100):
        ln.submit(srs, 0, -20000, dp, 50000); out.append(ln.collect())
    lanes_ok.append(len(set(out)) == 1 and out[0]); ln.close()
th = [threading.Thread(target=mwork, args=(k,)) for k in range(4)]
[t.start() for t in th]; [t.join() for t in th]
print("4 threads x 100 MSMs: all equal:", len(set(lanes_ok)) == 1 and lanes_ok[0] not in (False,))
