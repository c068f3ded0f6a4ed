"""Shared helpers for the parity tests: synthetic circuits in the encodings both sides take."""
import os
import random

import numpy as np

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
NCPU = os.cpu_count() or 1


def fr_bytes(vals):
    return np.frombuffer(b"".join((int(v) % R).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32).copy() \
        if len(vals) else np.zeros((0, 32), np.uint8)


def rand_fr_array(rng: np.random.Generator, n: int) -> np.ndarray:
    """n uniform-ish canonical Fr as uint8 [n, 32] without python big-int loops: top byte < 0x73."""
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] = rng.integers(0, 0x73, size=n, dtype=np.uint8)
    return a


def circuit_arrays(ref, pyrng: random.Random, n: int, Q: int):
    """rndCircuit (test/Test/Reference.hs:125-169) as (python lists, encoded arrays)."""
    circ, asg = ref.rnd_circuit(pyrng, n, Q)
    wL, wR, wO, cs = circ
    aL, aR, aO = asg
    enc = dict(wL=fr_bytes([v for r_ in wL for v in r_]), wR=fr_bytes([v for r_ in wR for v in r_]),
               wO=fr_bytes([v for r_ in wO for v in r_]), cs=fr_bytes(cs), aL=fr_bytes(aL), aR=fr_bytes(aR), aO=fr_bytes(aO))
    return circ, asg, enc


def big_circuit(seed: int, n: int, Q: int, orc):
    """Same generator at scale, numpy-side: aL, aR uniform, aO = aL*aR (via the oracle's Fr mul is too slow
    for 2^18, so aO is computed with python ints in chunks), weights with one all-ones row each."""
    rng = np.random.default_rng(seed)
    aL = rand_fr_array(rng, n)
    aR = rand_fr_array(rng, n)
    la = [int.from_bytes(aL[i].tobytes(), "little") for i in range(n)]
    lb = [int.from_bytes(aR[i].tobytes(), "little") for i in range(n)]
    lo = [a * b % R for a, b in zip(la, lb)]
    aO = fr_bytes(lo)
    rows = rng.integers(0, Q, size=3)
    one = (1).to_bytes(32, "little")
    W = []
    for r_ in rows:
        w = np.zeros((Q, n, 32), np.uint8)
        w[r_, :, :] = np.frombuffer(one, np.uint8)
        W.append(w.reshape(-1, 32))
    sums = [sum(la) % R, sum(lb) % R, sum(lo) % R]
    cs = [0] * Q
    for k, r_ in enumerate(rows):
        cs[int(r_)] = (cs[int(r_)] + sums[k]) % R
    return dict(wL=W[0], wR=W[1], wO=W[2], cs=fr_bytes(cs), aL=aL, aR=aR, aO=aO,
                rows=[int(r_) for r_ in rows], ints=(la, lb, lo), cs_ints=cs)
