"""Shared helpers for the parity tests: synthetic circuits in the encodings both sides take."""
import os
import random

import numpy as np

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
NCPU = os.cpu_count() or 1


from sonic_amd.workload import big_circuit, fr_bytes, rand_fr_array  # noqa: E402,F401  (the generators live with the product: bench.py and the tools use them too)


def circuit_arrays(ref, pyrng: random.Random, n: int, Q: int):
    """rndCircuit (test/Test/Reference.hs:125-169) as (python lists, encoded arrays)."""
    circ, asg = ref.rnd_circuit(pyrng, n, Q)
    wL, wR, wO, cs = circ
    aL, aR, aO = asg
    enc = dict(wL=fr_bytes([v for r_ in wL for v in r_]), wR=fr_bytes([v for r_ in wR for v in r_]),
               wO=fr_bytes([v for r_ in wO for v in r_]), cs=fr_bytes(cs), aL=fr_bytes(aL), aR=fr_bytes(aR), aO=fr_bytes(aO))
    return circ, asg, enc
