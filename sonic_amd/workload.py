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


def wt_quotient_scalars(sonic, seed: int, n: int, Q: int = 2):
    """The scalars the protocol itself hands to its two largest kinds of MSM, for a rndCircuit statement (bench.py `msm_protocol_shaped`,
    SURVEY 8d "protocol-shaped scalar set"):

      * the 7n + 8 quotient coefficients of W_t = openPoly(t(X,y), z) (src/Sonic/Protocol.hs:81, CommitmentScheme.hs:43-48), exponents
        [-4n-8, 3n-1], with t(X,y) = r(X,1) (r(X,y) + s(X,y)) - k(y) (Constraints.hs:56-68) -- dense and pseudo-random;
      * the 3n + 1 coefficients of s(X,y) (Constraints.hs:34-53 with Y := y), exponents [-n, 2n] -- n copies of one value, n of
        another, n distinct ones and a zero: the shape that makes heavy buckets.

    Built on the host: python integers for the O(n) parts, the product's own transform (sonic_poly_mul_fr) for the one product.
    Returns (uint8 [7n+8, 32], uint8 [3n+1, 32])."""
    from . import _lib
    c = big_circuit(seed, n, Q)
    la, lb, lo = c["ints"]
    rows = c["rows"]                      # the all-ones row of wL, wR, wO (0-based): q = row + 1
    rng = np.random.default_rng(seed + 1)
    tr = [int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1 for _ in range(6)]
    blind, y, z = tr[:4], tr[4], tr[5]
    yi = pow(y, -1, R)
    # y^e for e in [-2n-4, 2n+Q+1]
    top = 2 * n + Q + 2
    pw_pos = [1] * (top + 1)
    for e in range(1, top + 1):
        pw_pos[e] = pw_pos[e - 1] * y % R
    pw_neg = [1] * (2 * n + 5)
    for e in range(1, 2 * n + 5):
        pw_neg[e] = pw_neg[e - 1] * yi % R
    ypow = lambda e: pw_pos[e] if e >= 0 else pw_neg[-e]        # noqa: E731
    # r(X,1) over [-2n-4, n]  (Constraints.hs:23-31 + the blinders of Protocol.hs:58-62)
    r_lo = -2 * n - 4
    r1 = [0] * (3 * n + 5)
    for i in range(1, n + 1):
        r1[i - r_lo] = la[i - 1]
        r1[-i - r_lo] = lb[i - 1]
        r1[-i - n - r_lo] = lo[i - 1]
    for i in range(1, 5):
        r1[-2 * n - i - r_lo] = blind[i - 1]
    # r(X,y) + s(X,y) over [-2n-4, 2n]
    b_len = 4 * n + 5
    rs = [0] * b_len
    for k, v in enumerate(r1):
        if v:
            rs[k] = v * ypow(r_lo + k) % R
    uL, vR, wO = ypow(n + rows[0] + 1), ypow(n + rows[1] + 1), ypow(n + rows[2] + 1)
    sy = [0] * (3 * n + 1)                # s(X,y) over [-n, 2n]
    for i in range(1, n + 1):
        sy[-i + n] = uL
        sy[i + n] = vR
        sy[i + n + n] = (wO - ypow(i) - ypow(-i)) % R
    for k, v in enumerate(sy):
        rs[k - n - r_lo] = (rs[k - n - r_lo] + v) % R
    a = fr_bytes(r1)
    b = fr_bytes(rs)
    out = np.zeros((len(r1) + len(rs) - 1, 32), np.uint8)
    _lib.check(_lib.lib().sonic_poly_mul_fr(a.ctypes.data, len(r1), b.ctypes.data, len(rs), out.ctypes.data))
    t = [int.from_bytes(out[k].tobytes(), "little") for k in range(out.shape[0])]       # exponents [-4n-8, 3n]
    t_lo = -4 * n - 8
    ky = sum(c["cs_ints"][q] * ypow(n + q + 1) for q in range(Q)) % R
    t[-t_lo] = (t[-t_lo] - ky) % R
    if t[-t_lo] != 0:
        raise RuntimeError("wt_quotient_scalars: t(X,y) has a constant term: the synthetic circuit is not satisfied")
    # (t(X) - t(z)) / (X - z): synthetic division of X^m t(X) - t(z) X^m, m = -t_lo (CommitmentScheme.hs:43-44)
    zi = pow(z, -1, R)
    tz, p = 0, pow(zi, -t_lo, R)
    for v in t:
        tz = (tz + v * p) % R
        p = p * z % R
    g = list(t)
    g[-t_lo] = (g[-t_lo] - tz) % R
    q = [0] * (len(g) - 1)
    carry = 0
    for k in range(len(g) - 1, 0, -1):
        carry = (g[k] + z * carry) % R
        q[k - 1] = carry
    if (g[0] + z * carry) % R != 0:
        raise RuntimeError("wt_quotient_scalars: the division by X - z left a remainder")
    return fr_bytes(q), fr_bytes(sy)
