"""Synthetic workloads at scale: the reference's `rndCircuit` generator (test/Test/Reference.hs:125-169) restated with numpy so
that circuits of 2^18 .. 2^20 gates are made in seconds -- used by bench.py, the tools and the parity tests alike.  Everything is
drawn from numpy's PCG64 keyed by `seed`; values are canonical Fr encodings (uint8 [n, 32], little-endian)."""
from __future__ import annotations

import numpy as np

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr_bytes(vals) -> np.ndarray:
    return np.frombuffer(b"".join((int(v) % R).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32).copy() \
        if len(vals) else np.zeros((0, 32), np.uint8)


def rand_fr_array(rng: np.random.Generator, n: int) -> np.ndarray:
    """n uniform-ish canonical Fr as uint8 [n, 32] without python big-int loops: top byte < 0x73."""
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] = rng.integers(0, 0x73, size=n, dtype=np.uint8)
    return a


def big_circuit(seed: int, n: int, Q: int, orc=None):
    """rndCircuit at scale: aL, aR uniform, aO = aL * aR (python integers), weights with one all-ones row each
    (test/Test/Reference.hs:141-155), cs = wL aL + wR aR + wO aO (:138).  `orc` is unused (kept for old call sites)."""
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
