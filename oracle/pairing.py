"""CPU oracle (test infrastructure): BLS12-381 G2, the ate pairing and the reference's verifier.

TEST INFRASTRUCTURE ONLY -- nothing under sonic_amd/ imports this.

Restates, with python big integers, the three verifier functions of the reference so that its only
end-to-end test -- `verify srs circuit proof y z yzs` accepts what `prove` produced
(test/Test/Protocol.hs:14-23, test/Test/Signature.hs:20-36, test/Test/CommitmentScheme.hs:25-96) -- can be
run on proofs made by the HIP path, with real pairings instead of the known-trapdoor identity:

    pcV        src/Sonic/CommitmentScheme.hs:51-68     e(W, h^{alpha x}) e(g^v W^{-z}, h^alpha) == e(F, h^{x^{-d+max}})
    hscVerify  src/Sonic/Signature.hs:74-90
    verify     src/Sonic/Protocol.hs:111-130

The pairing itself lives in pairing-1.0.0 (un-vendored, absent).  Only equalities of pairing products are
ever tested, so any bilinear non-degenerate pairing on (G1, G2) accepts exactly the same proofs; this file
uses the textbook Miller loop over |x| = 0xd201000000010000 on the untwisted curve in Fq12 =
Fq[w]/(w^12 - 2 w^6 + 2), followed by the full final exponentiation (q^12 - 1)/r.  `selfcheck()` verifies
the G2 generator (on the twist, order r) and bilinearity numerically.
"""
from __future__ import annotations

from .sonic_ref import (G1_GEN, INF, Q, R, SRS as G1SRS, eval_x, eval_y, g1_add, g1_mul, g1_neg, k_poly, lp_eval, s_poly)

ATE_LOOP = 0xD201000000010000

# ---- Fq2 = Fq[u]/(u^2 + 1) as pairs ------------------------------------------------------------
def f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2_neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2_sqr(a): return f2_mul(a, a)
def f2_scalar(a, k): return (a[0] * k % Q, a[1] * k % Q)
def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * d % Q, (-a[1]) * d % Q)
F2_ZERO, F2_ONE = (0, 0), (1, 0)
B2 = (4, 4)                                   # twist: y^2 = x^3 + 4(u + 1)

G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)


def g2_on_curve(p):
    if p is INF:
        return True
    x, y = p
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B2)) == F2_ZERO


def g2_add(p, q):
    if p is INF:
        return q
    if q is INF:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if f2_add(y1, y2) == F2_ZERO:
            return INF
        lam = f2_mul(f2_scalar(f2_sqr(x1), 3), f2_inv(f2_scalar(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_mul(p, k):
    k %= R
    acc = INF
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, p)
    return acc


# ---- Fq12 = Fq[w]/(w^12 - 2 w^6 + 2) as 12 coefficients ------------------------------------------
def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):                # w^12 = 2 w^6 - 2
        c = t[k]
        if c:
            t[k - 6] += 2 * c
            t[k - 12] -= 2 * c
    return [c % Q for c in t[:12]]


F12_ONE = [1] + [0] * 11


def f12_pow(a, e):
    acc = F12_ONE
    for bit in bin(e)[2:]:
        acc = f12_mul(acc, acc)
        if bit == "1":
            acc = f12_mul(acc, a)
    return acc


def f12_inv(a):
    """extended Euclid on polynomials over Fq (degree < 12), modulus w^12 - 2 w^6 + 2"""
    def deg(p):
        d = len(p) - 1
        while d and p[d] == 0:
            d -= 1
        return d

    def divmod_poly(a_, b_):
        a_ = list(a_)
        o = [0] * len(a_)
        db = deg(b_)
        inv_lead = pow(b_[db], -1, Q)
        for i in range(deg(a_) - db, -1, -1):
            c = a_[db + i] * inv_lead % Q
            o[i] = c
            for j in range(db + 1):
                a_[i + j] = (a_[i + j] - c * b_[j]) % Q
        return o, a_

    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [2, 0, 0, 0, 0, 0, Q - 2, 0, 0, 0, 0, 0, 1]
    while deg(low):
        r_, _ = divmod_poly(high, low)
        r_ += [0] * (13 - len(r_))
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r_[j]) % Q
                new[i + j] = (new[i + j] - low[i] * r_[j]) % Q
        lm, low, hm, high = nm, new, lm, low
    c = pow(low[0], -1, Q)
    return [x * c % Q for x in lm[:12]]


def _f2_to_f12(a):
    """c0 + c1 u  with u = w^6 - 1   (w^6 = 1 + u since w^12 - 2w^6 + 2 = 0  <=>  (w^6 - 1)^2 = -1)"""
    return [(a[0] - a[1]) % Q, 0, 0, 0, 0, 0, a[1] % Q, 0, 0, 0, 0, 0]


W2_INV = f12_inv([0, 0, 1] + [0] * 9)
W3_INV = f12_inv([0, 0, 0, 1] + [0] * 8)


def untwist(qpt):
    """G2 point on the twist y^2 = x^3 + 4(u+1)  ->  E(Fq12): y^2 = x^3 + 4  via (x / w^2, y / w^3)"""
    return (f12_mul(_f2_to_f12(qpt[0]), W2_INV), f12_mul(_f2_to_f12(qpt[1]), W3_INV))


def _f12_add(a, b): return [(x + y) % Q for x, y in zip(a, b)]
def _f12_sub(a, b): return [(x - y) % Q for x, y in zip(a, b)]
def _f12_scalar(a, k): return [x * k % Q for x in a]


def _line(p1, p2, t):
    """value at t of the line through p1, p2 (tangent if equal) on E(Fq12)"""
    (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
    if x1 != x2:
        m = f12_mul(_f12_sub(y2, y1), f12_inv(_f12_sub(x2, x1)))
        return _f12_sub(f12_mul(m, _f12_sub(xt, x1)), _f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(_f12_scalar(f12_mul(x1, x1), 3), f12_inv(_f12_scalar(y1, 2)))
        return _f12_sub(f12_mul(m, _f12_sub(xt, x1)), _f12_sub(yt, y1))
    return _f12_sub(xt, x1)


def _e12_add(p1, p2):
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2 and y1 == y2:
        m = f12_mul(_f12_scalar(f12_mul(x1, x1), 3), f12_inv(_f12_scalar(y1, 2)))
    else:
        m = f12_mul(_f12_sub(y2, y1), f12_inv(_f12_sub(x2, x1)))
    x3 = _f12_sub(_f12_sub(f12_mul(m, m), x1), x2)
    return (x3, _f12_sub(f12_mul(m, _f12_sub(x1, x3)), y1))


def miller_loop(p, q):
    """f_{|x|, Q}(P) without the final exponentiation; 1 if either point is infinity"""
    if p is INF or q is INF:
        return F12_ONE
    Qp = untwist(q)
    Pt = ([p[0]] + [0] * 11, [p[1]] + [0] * 11)
    Rr, f = Qp, F12_ONE
    for bit in bin(ATE_LOOP)[3:]:
        f = f12_mul(f12_mul(f, f), _line(Rr, Rr, Pt))
        Rr = _e12_add(Rr, Rr)
        if bit == "1":
            f = f12_mul(f, _line(Rr, Qp, Pt))
            Rr = _e12_add(Rr, Qp)
    return f


FINAL_EXP = (Q ** 12 - 1) // R


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 with one shared final exponentiation"""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(p, q))
    return f12_pow(f, FINAL_EXP) == F12_ONE


def pairing(p, q):
    return f12_pow(miller_loop(p, q), FINAL_EXP)


# ---- the verifier's half of the SRS (SRS.hs:35-36,40-41) ----------------------------------------
class SRS(G1SRS):
    """adds the G2 vectors the verifier reads: hNegativeX, hPositiveX, hPositiveAlphaX"""

    def _h(self, name, k, length, scalar):
        if not (0 <= k < length):
            raise IndexError(f"{name} is not long enough: {k} >= {length}")   # CommitmentScheme.hs:70-73
        key = (name, k)
        if key not in self._cache:
            self._cache[key] = g2_mul(G2_GEN, scalar)
        return self._cache[key]

    def hNegativeX(self, k):          # SRS.hs:35
        return self._h("hNegativeX", k, self.d, pow(self.x_inv, k + 1, R))

    def hPositiveX(self, k):          # SRS.hs:36
        return self._h("hPositiveX", k, self.d + 1, pow(self.x, k, R))

    def hPositiveAlphaX(self, k):     # SRS.hs:41
        return self._h("hPositiveAlphaX", k, self.d + 1, self.alpha * pow(self.x, k, R) % R)


def pc_v(srs: SRS, maxm: int, commitment, z: int, opening) -> bool:
    """pcV (CommitmentScheme.hs:51-68):  eA <> eB == eC"""
    v, w = opening
    difference = -srs.d + maxm
    hxi = srs.hPositiveX(difference) if difference >= 0 else srs.hNegativeX(abs(difference) - 1)
    left_b = g1_add(g1_mul(G1_GEN, v), g1_mul(w, (-z) % R))
    return pairing_product_is_one([(w, srs.hPositiveAlphaX(1)), (left_b, srs.hPositiveAlphaX(0)), (g1_neg(commitment), hxi)])


def hsc_verify(srs: SRS, sXY, yzs, hsc) -> bool:
    """hscVerify (Signature.hs:80-90)"""
    sv = lp_eval(eval_y(hsc["hscV"], sXY), hsc["hscU"])
    ok = True
    for (yi, zi), (ci, (si, wi)), (sip, wip, qi) in zip(yzs, hsc["hscS"], hsc["hscW"]):
        ok = ok and pc_v(srs, srs.d, ci, zi, (si, wi)) and pc_v(srs, srs.d, ci, hsc["hscU"], (sip, wip)) \
            and pc_v(srs, srs.d, hsc["hscC"], yi, (sip, qi))
    return pc_v(srs, srs.d, hsc["hscC"], hsc["hscV"], (sv, hsc["hscQv"])) and ok


def verify(srs: SRS, circuit, proof, y: int, z: int, yzs) -> bool:
    """verify (Protocol.hs:111-130)"""
    wL, wR, wO, cs = circuit
    n = len(wL[0])
    kY = k_poly(cs, n)
    sXY = s_poly(wL, wR, wO)
    t = (proof["prA"] * (proof["prB"] + proof["prS"]) - lp_eval(kY, y)) % R
    checks = [hsc_verify(srs, sXY, yzs, proof["prHscProof"]),
              pc_v(srs, n, proof["prR"], z, (proof["prA"], proof["prWa"])),
              pc_v(srs, n, proof["prR"], y * z % R, (proof["prB"], proof["prWb"])),
              pc_v(srs, srs.d, proof["prT"], z, (t, proof["prWt"]))]
    return all(checks)


def proof_from_bytes(b: bytes, Q_lin: int):
    """inverse of sonic_ref.proof_to_bytes"""
    from .sonic_ref import g1_from_bytes
    pos = 0

    def g():
        nonlocal pos
        v = g1_from_bytes(b[pos:pos + 96]); pos += 96
        return v

    def f():
        nonlocal pos
        v = int.from_bytes(b[pos:pos + 32], "little"); pos += 32
        return v

    pr = {"prR": g(), "prT": g(), "prA": f(), "prWa": g(), "prB": f(), "prWb": g(), "prWt": g(), "prS": f()}
    hscS = []
    for _ in range(Q_lin):
        cm, sj, wj = g(), f(), g()
        hscS.append((cm, (sj, wj)))
    hscW = []
    for _ in range(Q_lin):
        sjp, wjp, qj = f(), g(), g()
        hscW.append((sjp, wjp, qj))
    qv, c, u, v = g(), g(), f(), f()
    assert pos == len(b)
    pr["prHscProof"] = {"hscS": hscS, "hscW": hscW, "hscQv": qv, "hscC": c, "hscU": u, "hscV": v}
    return pr


def selfcheck():
    assert g2_on_curve(G2_GEN)
    assert g2_mul(G2_GEN, R - 1) == (G2_GEN[0], f2_neg(G2_GEN[1]))          # order r
    a, b = 0x1234567, 0x7654321
    e = pairing(G1_GEN, G2_GEN)
    assert e != F12_ONE
    assert pairing(g1_mul(G1_GEN, a), g2_mul(G2_GEN, b)) == f12_pow(e, a * b)
    assert pairing_product_is_one([(g1_mul(G1_GEN, a), G2_GEN), (g1_neg(G1_GEN), g2_mul(G2_GEN, a))])
    return True


if __name__ == "__main__":
    import time
    t0 = time.time()
    print("selfcheck", selfcheck(), "%.1fs" % (time.time() - t0))
