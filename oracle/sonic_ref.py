"""CPU oracle (1 of 2): literal big-integer restatement of the sdiehl/sonic prover path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sonic_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use ``oracle/``.

PARITY UNPINNED at the byte level: the reference (Haskell, /root/reference) holds no golden
vectors or known-answer tests for this path (every test is a randomized accept/reject
property, test/Test/*.hs) and cannot be built here (no ghc/cabal/stack, and all of its
arithmetic lives in un-vendored packages: pairing-1.0.0, elliptic-curve-0.3.0,
galois-field-1.0.1@b59ecd8, poly-0.4.0.0@0ef404b, semirings-0.5.3, bulletproofs-1.1.0;
stack.yaml:5-14).  What pins this restatement instead:
  * Fr, Fq and E(Fq): y^2 = x^3 + 4 are standard objects; an MSM has one value and an affine
    point one canonical encoding.  `selfcheck()` verifies the curve constants numerically
    (q, r prime; generator on curve; r*G = O; 2-adicity of Fr).
  * the reference's own acceptance properties restated pairing-free through the known-trapdoor
    "exponent oracle" (`pcv_logs`, `verify_exponent`): tests build the SRS from known x, alpha
    (test/Test/Protocol.hs:21), so every commitment F has a known discrete log and
    pcV's pairing equation e(W,h^{ax}) e(g^v W^{-z},h^a) = e(F,h^{x^{-d+max}})
    (src/Sonic/CommitmentScheme.hs:58-68) is equivalent to an identity in Fr.
  * two public third-party known answers that need no hashing, EIP-2537's doubles of the G1 and G2
    generators (tests/golden/eip2537_kat.json): generators and group laws are the public ones.

Each function cites the reference lines it follows.  Polynomials are kept in the reference's own
shape: a sparse Laurent polynomial is a dict {exponent: coeff != 0} (poly's normalised VLaurent),
a bivariate one is {x_exponent: {y_exponent: coeff}} (BiVLaurent, src/Sonic/Utils.hs:15).
MSMs are the reference's left folds of  acc <> (P `mul` v).
"""
from __future__ import annotations

# --------------------------------------------------------------------------------------
# BLS12-381 constants (pairing-1.0.0 `Data.Pairing.BLS12381`; standard values)
# --------------------------------------------------------------------------------------
Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G1_GEN = (G1_X, G1_Y)
INF = None  # point at infinity (`mempty`)


def fr(x: int) -> int:
    return x % R


def fr_inv(x: int) -> int:
    if x % R == 0:
        raise ZeroDivisionError("Fr inverse of zero")
    return pow(x, -1, R)


def fr_pow(x: int, e: int) -> int:
    """galois-field `pow` with possibly negative exponent (src/Sonic/Utils.hs:18 passes e<0)."""
    if e >= 0:
        return pow(x, e, R)
    return pow(fr_inv(x), -e, R)


# --------------------------------------------------------------------------------------
# G1 group law (elliptic-curve-0.3.0 `Curve`: `<>` = add, `mul`, `gen`, `mempty`)
# --------------------------------------------------------------------------------------
def g1_is_on_curve(p) -> bool:
    if p is INF:
        return True
    x, y = p
    return (y * y - x * x * x - 4) % Q == 0


def g1_neg(p):
    if p is INF:
        return INF
    return (p[0], (-p[1]) % Q)


def g1_add(p, q):
    """Affine chord-and-tangent addition."""
    if p is INF:
        return q
    if q is INF:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return INF
        lam = (3 * x1 * x1) * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    y3 = (lam * (x1 - x3) - y1) % Q
    return (x3, y3)


def _jac_double(X, Y, Z):
    if Y == 0:
        return (1, 1, 0)
    A = X * X % Q
    B = Y * Y % Q
    C = B * B % Q
    D = 2 * ((X + B) * (X + B) - A - C) % Q
    E = 3 * A % Q
    F = E * E % Q
    X3 = (F - 2 * D) % Q
    Y3 = (E * (D - X3) - 8 * C) % Q
    Z3 = 2 * Y * Z % Q
    return (X3, Y3, Z3)


def _jac_add_affine(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % Q
    U2 = x2 * Z1Z1 % Q
    S2 = y2 * Z1 * Z1Z1 % Q
    if U2 == X1:
        if S2 == Y1:
            return _jac_double(X1, Y1, Z1)
        return (1, 1, 0)
    H = (U2 - X1) % Q
    HH = H * H % Q
    I = 4 * HH % Q
    J = H * I % Q
    r = 2 * (S2 - Y1) % Q
    V = X1 * I % Q
    X3 = (r * r - J - 2 * V) % Q
    Y3 = (r * (V - X3) - 2 * Y1 * J) % Q
    Z3 = ((Z1 + H) * (Z1 + H) - Z1Z1 - HH) % Q
    return (X3, Y3, Z3)


def g1_mul(p, k: int):
    """`mul :: G1 -> Fr -> G1` (CommitmentScheme.hs:26-28,45-47).  MSB-first double-and-add in
    Jacobian coordinates; result normalised to the unique affine representative."""
    k %= R
    if p is INF or k == 0:
        return INF
    x2, y2 = p
    X, Y, Z = 1, 1, 0
    for bit in bin(k)[2:]:
        X, Y, Z = _jac_double(X, Y, Z)
        if bit == "1":
            X, Y, Z = _jac_add_affine(X, Y, Z, x2, y2)
    if Z == 0:
        return INF
    zi = pow(Z, -1, Q)
    zi2 = zi * zi % Q
    return (X * zi2 % Q, Y * zi2 * zi % Q)


# --------------------------------------------------------------------------------------
# Canonical byte encodings used at the C ABI (the reference defines none, Protocol.hs:38)
# --------------------------------------------------------------------------------------
def fr_to_bytes(x: int) -> bytes:
    return (x % R).to_bytes(32, "little")


def fr_from_bytes(b: bytes) -> int:
    v = int.from_bytes(b, "little")
    assert v < R
    return v


def g1_to_bytes(p) -> bytes:
    if p is INF:
        return bytes(96)
    return p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little")


def g1_from_bytes(b: bytes):
    assert len(b) == 96
    if b == bytes(96):
        return INF
    return (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:], "little"))


# --------------------------------------------------------------------------------------
# Sparse Laurent polynomials (poly-0.4.0.0 `Data.Poly.Sparse.Laurent`)
# --------------------------------------------------------------------------------------
def lp_norm(p: dict) -> dict:
    return {e: c % R for e, c in p.items() if c % R != 0}


def lp_add(a: dict, b: dict) -> dict:
    out = dict(a)
    for e, c in b.items():
        out[e] = (out.get(e, 0) + c) % R
    return lp_norm(out)


def lp_neg(a: dict) -> dict:
    return {e: (-c) % R for e, c in a.items()}


def lp_mul(a: dict, b: dict) -> dict:
    """Sparse convolution (`*` on VLaurent)."""
    out: dict = {}
    for e1, c1 in a.items():
        for e2, c2 in b.items():
            out[e1 + e2] = (out.get(e1 + e2, 0) + c1 * c2) % R
    return lp_norm(out)


def lp_eval(p: dict, x: int) -> int:
    """`eval` on a Laurent polynomial: sum c_e x^e, negative e through x^-1."""
    acc = 0
    for e, c in p.items():
        acc = (acc + c * fr_pow(x, e)) % R
    return acc


def lp_items(p: dict):
    """`GHC.Exts.toList`: (exponent, coeff) in increasing exponent order."""
    return sorted(p.items())


def lp_divide_linear(p: dict, z: int) -> dict:
    """`(f - f(z)) `divide` (X - z)` (CommitmentScheme.hs:44) for a Laurent f with f(z) = 0:
    multiply by X^-lo to get an ordinary polynomial, synthetic division, shift back."""
    p = lp_norm(p)
    if not p:
        return {}
    if z % R == 0:
        # X - 0 = X is a unit of the Laurent ring: the quotient is the shift by one, always exact.  (The shift-and-Horner
        # restatement below needs z != 0: X^-lo is only a unit at z != 0.)  openPoly reaches this with a polynomial without
        # negative exponents only -- `eval` at 0 of a negative power has already failed (lp_eval: 0^-1).
        return lp_norm({e - 1: c for e, c in p.items()})
    lo, hi = min(p), max(p)
    coeffs = [p.get(e, 0) for e in range(lo, hi + 1)]  # ascending, degree D = hi-lo
    D = hi - lo
    quot = [0] * D
    carry = 0
    for k in range(D, 0, -1):  # Horner from the top
        carry = (coeffs[k] + carry * z) % R
        quot[k - 1] = carry
    rem = (coeffs[0] + carry * z) % R
    if rem != 0:
        raise ValueError("fromJust: inexact division")  # CommitmentScheme.hs:44 `fromJust`
    return lp_norm({lo + k: quot[k] for k in range(D)})


# bivariate: {xexp: {yexp: coeff}}
def biv_norm(p: dict) -> dict:
    out = {}
    for ex, py in p.items():
        q = lp_norm(py)
        if q:
            out[ex] = q
    return out


def biv_add(a: dict, b: dict) -> dict:
    out = {ex: dict(py) for ex, py in a.items()}
    for ex, py in b.items():
        out[ex] = lp_add(out.get(ex, {}), py)
    return biv_norm(out)


def biv_mul(a: dict, b: dict) -> dict:
    out: dict = {}
    for e1, p1 in a.items():
        for e2, p2 in b.items():
            out[e1 + e2] = lp_add(out.get(e1 + e2, {}), lp_mul(p1, p2))
    return biv_norm(out)


# --- src/Sonic/Utils.hs ---------------------------------------------------------------
def eval_x(x: int, p: dict) -> dict:
    """evalX (Utils.hs:17-18): sum over X-terms of  pow x e `scale` coeff(Y)."""
    out: dict = {}
    for ex, py in lp_items(p):
        s = fr_pow(x, ex)
        out = lp_add(out, {ey: c * s % R for ey, c in py.items()})
    return out


def eval_y(y: int, p: dict) -> dict:
    """evalY (Utils.hs:20-21): evaluate every X-coefficient at Y := y, renormalise."""
    return lp_norm({ex: lp_eval(py, y) for ex, py in p.items()})


def from_x(p: dict) -> dict:
    """fromX (Utils.hs:23-24): coefficient c -> constant-in-Y polynomial."""
    return {ex: {0: c} for ex, c in p.items()}


def from_y(p: dict) -> dict:
    """fromY (Utils.hs:26-27): monomial 0."""
    return biv_norm({0: dict(p)})


# --- src/Sonic/Constraints.hs ---------------------------------------------------------
def r_poly(aL, aR, aO) -> dict:
    """rPoly (Constraints.hs:23-31)."""
    n = len(aL)
    out: dict = {}
    for i, (a, b, c) in enumerate(zip(aL, aR, aO), start=1):
        for e, v in ((i, a), (-i, b), (-i - n, c)):
            out[e] = lp_add(out.get(e, {}), {e: v % R})
    return biv_norm(out)


def s_poly(wL, wR, wO) -> dict:
    """sPoly (Constraints.hs:34-53).  n = length (head wL)."""
    n = len(wL[0])

    def xiY(i, xL):
        acc: dict = {}
        for q, row in enumerate(xL, start=1):
            acc = lp_add(acc, {q + n: row[i - 1] % R})
        return acc

    out: dict = {}
    for i in range(1, n + 1):
        ui = xiY(i, wL)
        vi = xiY(i, wR)
        wi = lp_add(lp_add({-i: R - 1}, {i: R - 1}), xiY(i, wO))
        for e, py in ((-i, ui), (i, vi), (i + n, wi)):
            out[e] = lp_add(out.get(e, {}), py)
    return biv_norm(out)


def k_poly(cs, n) -> dict:
    """kPoly (Constraints.hs:67-68): zip [n+1..] k."""
    return lp_norm({n + 1 + i: c for i, c in enumerate(cs)})


def t_poly(rXY: dict, sXY: dict, kY: dict) -> dict:
    """tPoly (Constraints.hs:56-65): r(X,1) * (r(X,Y)+s(X,Y)) - k(Y)."""
    rXYp = biv_add(rXY, sXY)
    rX1 = from_x(eval_y(1, rXY))
    k1Y = from_y(lp_neg(kY))
    return biv_add(biv_mul(rX1, rXYp), k1Y)


# --- src/Sonic/SRS.hs -----------------------------------------------------------------
class SRS:
    """SRS.new (SRS.hs:27-43).  Like the reference's boxed lazy vectors, elements are computed
    on first use.  G2 vectors are not modelled (never read by the prover)."""

    def __init__(self, d: int, x: int, alpha: int):
        self.d, self.x, self.alpha = d, x % R, alpha % R
        self.x_inv = fr_inv(x)
        self._cache: dict = {}

    def _get(self, name, k, length, scalar_fn):
        if not (0 <= k < length):
            raise IndexError(f"{name} is not long enough: {k} >= {length}")  # CommitmentScheme.hs:70-73
        key = (name, k)
        if key not in self._cache:
            self._cache[key] = g1_mul(G1_GEN, scalar_fn(k))
        return self._cache[key]

    def gNegativeX(self, k):  # SRS.hs:33  g^{x^{-(k+1)}}
        return self._get("gNegativeX", k, self.d, lambda k: pow(self.x_inv, k + 1, R))

    def gPositiveX(self, k):  # SRS.hs:34  g^{x^k}, k = 0..d
        return self._get("gPositiveX", k, self.d + 1, lambda k: pow(self.x, k, R))

    def gNegativeAlphaX(self, k):  # SRS.hs:37
        return self._get("gNegativeAlphaX", k, self.d, lambda k: self.alpha * pow(self.x_inv, k + 1, R) % R)

    def gPositiveAlphaX(self, k):  # SRS.hs:39  g^{alpha x^{k+1}} (g^alpha omitted)
        return self._get("gPositiveAlphaX", k, self.d, lambda k: self.alpha * pow(self.x, k + 1, R) % R)


# --- src/Sonic/CommitmentScheme.hs ----------------------------------------------------
def commit_poly(srs: SRS, maxm: int, fX: dict):
    """commitPoly (CommitmentScheme.hs:20-33)."""
    difference = srs.d - maxm
    xfX = sorted((e + difference, v) for e, v in lp_norm(fX).items())
    acc = INF
    for e, v in xfX:
        if e > 0:
            base = srs.gPositiveAlphaX(e - 1)
        else:
            base = srs.gNegativeAlphaX(abs(e) - 1)  # e == 0 -> index -1 -> panic
        acc = g1_add(acc, g1_mul(base, v))
    return acc


def open_poly(srs: SRS, z: int, fX: dict):
    """openPoly (CommitmentScheme.hs:36-48): (f(z), g^{(f(X)-f(z))/(X-z)})."""
    fz = lp_eval(fX, z)
    w_poly = lp_divide_linear(lp_add(fX, {0: (-fz) % R}), z)
    acc = INF
    for e, v in lp_items(w_poly):
        base = srs.gPositiveX(e) if e >= 0 else srs.gNegativeX(abs(e) - 1)
        acc = g1_add(acc, g1_mul(base, v))
    return fz, acc


def commit_log(srs: SRS, maxm: int, fX: dict) -> int:
    """dlog of commitPoly's result: alpha * x^{d-max} * f(x)."""
    return srs.alpha * fr_pow(srs.x, srs.d - maxm) % R * lp_eval(fX, srs.x) % R


def open_log(srs: SRS, z: int, fX: dict) -> int:
    """dlog of openPoly's W: (f(x) - f(z)) / (x - z)."""
    return (lp_eval(fX, srs.x) - lp_eval(fX, z)) % R * fr_inv(srs.x - z) % R


def pcv_logs(srs: SRS, maxm: int, F_log: int, z: int, v: int, W_log: int) -> bool:
    """pcV (CommitmentScheme.hs:51-68) through the known trapdoor.  With F = g^F', W = g^W':
    e(W,h^{ax}) e(g^v W^{-z}, h^a) == e(F, h^{x^{-d+max}})  <=>
    a*x*W' + a*(v - z*W') == F' * x^{-d+max}   in Fr."""
    a, x = srs.alpha, srs.x
    lhs = (a * x % R * W_log + a * ((v - z * W_log) % R)) % R
    rhs = F_log * fr_pow(x, -srs.d + maxm) % R
    return lhs == rhs


# --- src/Sonic/Signature.hs -----------------------------------------------------------
def hsc_prove(srs: SRS, sXY: dict, yzs, u: int, v: int):
    """hscProve (Signature.hs:38-72); u, v are the two `rnd` draws (:48, :60)."""
    hscS = []
    for yi, zi in yzs:  # :40-45
        sXy = eval_y(yi, sXY)
        cm = commit_poly(srs, srs.d, sXy)
        op = open_poly(srs, zi, sXy)
        hscS.append((cm, op))
    suX = eval_x(u, sXY)  # :51
    c = commit_poly(srs, srs.d, suX)  # :52
    hscW = []
    for yi, _zi in yzs:  # :53-57
        _, wjp = open_poly(srs, u, eval_y(yi, sXY))
        sjp, qj = open_poly(srs, yi, suX)
        hscW.append((sjp, wjp, qj))
    _, qv = open_poly(srs, v, suX)  # :63
    return {"hscS": hscS, "hscW": hscW, "hscQv": qv, "hscC": c, "hscU": u % R, "hscV": v % R}


# --- src/Sonic/Protocol.hs ------------------------------------------------------------
class DTooSmall(Exception):
    pass


def transcript_len(Q_lin: int) -> int:
    return 8 + 2 * Q_lin


def prove(srs: SRS, assignment, circuit, transcript):
    """prove (Protocol.hs:47-109).  `transcript` replaces the `rnd` draws, in draw order:
    cns[4] (:58), y (:66), z (:76), ys[m] (:84), zs[m] (:85), then hscProve's u, v."""
    aL, aR, aO = assignment
    wL, wR, wO, cs = circuit
    n = len(aL)
    m = len(wL)
    if srs.d < 7 * n:  # :54-55
        raise DTooSmall(f"Parameter d is not large enough: {srs.d} should be greater than {7 * n}")
    t = [v % R for v in transcript]
    assert len(t) == transcript_len(m)
    cns, y, z = t[0:4], t[4], t[5]
    ys, zs = t[6:6 + m], t[6 + m:6 + 2 * m]
    u, v = t[6 + 2 * m], t[7 + 2 * m]

    sumcXY = biv_norm({-(2 * n + i): {-(2 * n + i): c} for i, c in enumerate(cns, start=1)})  # :59-61
    polyRp = biv_add(r_poly(aL, aR, aO), sumcXY)  # :62
    rX1 = eval_y(1, polyRp)
    commitR = commit_poly(srs, n, rX1)  # :63
    kY = k_poly(cs, n)  # :69
    sXY = s_poly(wL, wR, wO)  # :70
    tXY = t_poly(polyRp, sXY, kY)  # :71
    tXy = eval_y(y, tXY)  # :72
    commitT = commit_poly(srs, srs.d, tXy)  # :73
    a, wa = open_poly(srs, z, rX1)  # :79
    b, wb = open_poly(srs, y * z % R, rX1)  # :80
    _, wt = open_poly(srs, z, tXy)  # :81
    szy = lp_eval(eval_y(y, sXY), z)  # :83
    yzs = list(zip(ys, zs))  # :86
    hsc = hsc_prove(srs, sXY, yzs, u, v)  # :87
    proof = {"prR": commitR, "prT": commitT, "prA": a, "prWa": wa, "prB": b, "prWb": wb,
             "prWt": wt, "prS": szy, "prHscProof": hsc}
    oracle = {"y": y, "z": z, "yzs": yzs}
    return proof, oracle


def proof_to_bytes(proof) -> bytes:
    """Canonical proof encoding: record order of Proof then HscProof."""
    h = proof["prHscProof"]
    out = [g1_to_bytes(proof["prR"]), g1_to_bytes(proof["prT"]), fr_to_bytes(proof["prA"]),
           g1_to_bytes(proof["prWa"]), fr_to_bytes(proof["prB"]), g1_to_bytes(proof["prWb"]),
           g1_to_bytes(proof["prWt"]), fr_to_bytes(proof["prS"])]
    for cm, (sj, wj) in h["hscS"]:
        out += [g1_to_bytes(cm), fr_to_bytes(sj), g1_to_bytes(wj)]
    for sjp, wjp, qj in h["hscW"]:
        out += [fr_to_bytes(sjp), g1_to_bytes(wjp), g1_to_bytes(qj)]
    out += [g1_to_bytes(h["hscQv"]), g1_to_bytes(h["hscC"]), fr_to_bytes(h["hscU"]), fr_to_bytes(h["hscV"])]
    return b"".join(out)


def verify_exponent(srs: SRS, circuit, assignment, transcript, proof) -> bool:
    """verify (Protocol.hs:111-130) + hscVerify (Signature.hs:80-90) restated through the known
    trapdoor: every pcV becomes (i) group element == g^{claimed log} and (ii) `pcv_logs`.
    The claimed logs are recomputed from the witness polynomials, which the honest-prover test
    setting has (test/Test/Protocol.hs:14-23 runs prove and verify in one process)."""
    aL, aR, aO = assignment
    wL, wR, wO, cs = circuit
    n, m = len(aL), len(wL)
    t = [v % R for v in transcript]
    cns, y, z = t[0:4], t[4], t[5]
    ys, zs = t[6:6 + m], t[6 + m:6 + 2 * m]
    u, v = t[6 + 2 * m], t[7 + 2 * m]
    sumcXY = biv_norm({-(2 * n + i): {-(2 * n + i): c} for i, c in enumerate(cns, start=1)})
    polyRp = biv_add(r_poly(aL, aR, aO), sumcXY)
    rX1 = eval_y(1, polyRp)
    sXY = s_poly(wL, wR, wO)
    kY = k_poly(cs, n)
    tXy = eval_y(y, t_poly(polyRp, sXY, kY))
    ok = True

    def chk(maxm, fX, F, zz, val, W):
        Fl, Wl = commit_log(srs, maxm, fX), open_log(srs, zz, fX)
        return (F == g1_mul(G1_GEN, Fl) and W == g1_mul(G1_GEN, Wl) and pcv_logs(srs, maxm, Fl, zz, val, Wl))

    tval = (proof["prA"] * (proof["prB"] + proof["prS"]) - lp_eval(kY, y)) % R  # Protocol.hs:120
    ok &= chk(n, rX1, proof["prR"], z, proof["prA"], proof["prWa"])  # :123
    ok &= chk(n, rX1, proof["prR"], y * z % R, proof["prB"], proof["prWb"])  # :124
    ok &= chk(srs.d, tXy, proof["prT"], z, tval, proof["prWt"])  # :125
    h = proof["prHscProof"]
    suX = eval_x(h["hscU"], sXY)
    sv = lp_eval(eval_y(h["hscV"], sXY), h["hscU"])  # Signature.hs:81
    for (yi, zi), (ci, (si, wi)), (sip, wip, qi) in zip(zip(ys, zs), h["hscS"], h["hscW"]):
        sXy = eval_y(yi, sXY)
        ok &= chk(srs.d, sXy, ci, zi, si, wi)  # :84
        ok &= chk(srs.d, sXy, ci, h["hscU"], sip, wip)  # :85
        ok &= chk(srs.d, suX, h["hscC"], yi, sip, qi)  # :86
    ok &= chk(srs.d, suX, h["hscC"], h["hscV"], sv, h["hscQv"])  # :89
    return bool(ok)


# --------------------------------------------------------------------------------------
# Circuits: the reference's fixed examples and generator (test/Test/Reference.hs)
# --------------------------------------------------------------------------------------

# ---- opt-in Fiat-Shamir transcript (restates sonic_amd/csrc/fs.hpp) -----------------------------------------
# The reference draws its challenges with `rnd`: y after R (Protocol.hs:66), z after T (:76), y_j / z_j after the openings
# (:84-85), u after the S_j (Signature.hs:48), v after C and the W'_j, Q_j (:60).  Here each draw is SHA-256 of what precedes
# it, in that order.  hashlib stands in for the product's own SHA-256.
import hashlib as _hashlib


def _le64(v: int) -> bytes:
    return int(v).to_bytes(8, "little")


def fs_wide(d0: bytes, d1: bytes) -> int:
    return int.from_bytes(d0 + d1, "little") % R


def fs_circuit_digest(circuit) -> bytes:
    wL, wR, wO, cs = circuit
    n, m = len(wL[0]), len(wL)
    h = _hashlib.sha256(b"sonic-hip/circuit/v1" + _le64(n) + _le64(m))
    for w in (wL, wR, wO):
        for row in w:
            for v in row:
                h.update(fr_to_bytes(v))
    for c in cs:
        h.update(fr_to_bytes(c))
    return h.digest()


def fs_srs_id(srs: SRS) -> bytes:
    """what ties a transcript to ONE structured reference string of a given d: g^x, g^{alpha x}, g^{1/x}, g^{alpha/x}
    (gPositiveX[1], gPositiveAlphaX[0], gNegativeX[0], gNegativeAlphaX[0]; SRS.hs:33-39) determine x and alpha"""
    pts = [srs.gPositiveX(1), srs.gPositiveAlphaX(0), srs.gNegativeX(0), srs.gNegativeAlphaX(0)]
    return _hashlib.sha256(b"sonic-hip/srs/v1" + _le64(srs.d) + b"".join(g1_to_bytes(p) for p in pts)).digest()


def fs_witness_digest(assignment) -> bytes:
    aL, aR, aO = assignment
    h = _hashlib.sha256(b"sonic-hip/witness/v1")
    for vec in (aL, aR, aO):
        for v in vec:
            h.update(fr_to_bytes(v))
    return h.digest()


def fs_blinders(seed: bytes, digest: bytes, srs_id: bytes, witness_digest: bytes):
    """the four blinders (Protocol.hs:58) from the prover's secret seed AND the statement, the SRS and the witness (in the manner of
    RFC 6979): the same seed on another assignment gives unrelated blinders, so R1 - R2 never is an unblinded commitment"""
    out = []
    for i in range(4):
        d = [_hashlib.sha256(b"sonic-hip/blinder/v2" + seed + digest + srs_id + witness_digest + i.to_bytes(4, "little") + bytes([half])).digest()
             for half in (0, 1)]
        out.append(fs_wide(d[0], d[1]))
    return out


class FsTranscript:
    def __init__(self, n: int, m: int, d: int, digest: bytes, srs_id: bytes):
        self.st = _hashlib.sha256(b"sonic-hip/fs/v2" + _le64(n) + _le64(m) + _le64(d) + digest + srs_id).digest()

    def absorb(self, label: bytes, data: bytes):
        self.st = _hashlib.sha256(self.st + label + data).digest()

    def challenge(self, label: bytes, i: int) -> int:
        d = [_hashlib.sha256(self.st + label + i.to_bytes(4, "little") + bytes([half])).digest() for half in (0, 1)]
        return fs_wide(d[0], d[1]) or 1


def fs_challenges_of_proof(n: int, m: int, d: int, digest: bytes, srs_id: bytes, pb: bytes):
    """y, z, [y_j], [z_j], u, v as a proof's canonical bytes determine them"""
    t = FsTranscript(n, m, d, digest, srs_id)
    t.absorb(b"R", pb[0:96])
    y = t.challenge(b"y", 0)
    t.absorb(b"T", pb[96:192])
    z = t.challenge(b"z", 0)
    t.absorb(b"open", pb[192:576])
    ys = [t.challenge(b"yj", j) for j in range(m)]
    zs = [t.challenge(b"zj", j) for j in range(m)]
    hs, hw = 576, 576 + 224 * m
    t.absorb(b"hscS", pb[hs:hw])
    u = t.challenge(b"u", 0)
    qv = hw + 224 * m
    t.absorb(b"hscW", pb[qv + 96:qv + 192] + pb[hw:qv])
    v = t.challenge(b"v", 0)
    return y, z, ys, zs, u, v


def prove_fs(srs: SRS, assignment, circuit, seed: bytes):
    """prove with every draw replaced by the hash of what precedes it.  Every proof element depends only on draws made before
    it (that is what lets the reference draw them on the way), so proving with the draws known so far and re-deriving the next
    one from the bytes reaches the fixed point after one round per draw site; the literal `prove` above does the proving."""
    aL = assignment[0]
    n, m = len(aL), len(circuit[0])
    digest, sid = fs_circuit_digest(circuit), fs_srs_id(srs)
    tr = fs_blinders(seed, digest, sid, fs_witness_digest(assignment)) + [1] * (4 + 2 * m)
    for _ in range(6):
        proof, _o = prove(srs, assignment, circuit, tr)
        y, z, ys, zs, u, v = fs_challenges_of_proof(n, m, srs.d, digest, sid, proof_to_bytes(proof))
        tr = tr[:4] + [y, z] + ys + zs + [u, v]
    proof, oracle = prove(srs, assignment, circuit, tr)
    assert fs_challenges_of_proof(n, m, srs.d, digest, sid, proof_to_bytes(proof)) == (y, z, ys, zs, u, v)
    return proof, oracle, tr

def arith_circuit_example1():
    """arithCircuitExample1 (test/Test/Reference.hs:38-50): n=1, Q=2."""
    wL, wR, wO = [[1], [0]], [[0], [1]], [[0], [0]]
    cs = [7 + 3, 2 + 10]
    aL, aR = [10], [12]
    aO = [a * b % R for a, b in zip(aL, aR)]
    return (wL, wR, wO, cs), (aL, aR, aO)


def arith_circuit_example2(z: int):
    """arithCircuitExample2 (test/Test/Reference.hs:65-90) == examples/Main.hs:38-63: n=2, Q=5."""
    wL = [[0, 0], [1, 0], [0, 1], [0, 0], [0, 0]]
    wR = [[0, 0], [0, 0], [0, 0], [1, 0], [0, 1]]
    wO = [[1, R - 1], [0, 0], [0, 0], [0, 0], [0, 0]]
    cs = [0, (4 - z) % R, (9 - z) % R, (9 - z) % R, (4 - z) % R]
    aL = [(4 - z) % R, (9 - z) % R]
    aR = [(9 - z) % R, (4 - z) % R]
    aO = [a * b % R for a, b in zip(aL, aR)]
    return (wL, wR, wO, cs), (aL, aR, aO)


def rnd_circuit(rng, n: int, m: int):
    """rndCircuit / arithCircuitGen / arithAssignmentGen (test/Test/Reference.hs:125-169):
    aL, aR uniform, aO = aL*aR; each of wL, wR, wO = (m-1) zero rows with one all-ones row
    inserted at i ~ U{0..m} (insertAt clamps i = m to the end, :143-145,154-155);
    cs = wL.aL + wR.aR + wO.aO (:138,159-162)."""
    aL = [rng.randrange(R) for _ in range(n)]
    aR = [rng.randrange(R) for _ in range(n)]
    aO = [a * b % R for a, b in zip(aL, aR)]

    def gen_vec():
        i = rng.randint(0, m)
        rows = [[0] * n for _ in range(m - 1)]
        return rows[:i] + [[1] * n] + rows[i:]

    wL, wR, wO = gen_vec(), gen_vec(), gen_vec()

    def tdot(v, mat):
        return [sum(a * b for a, b in zip(v, row)) % R for row in mat]

    cs = [(a + b + c) % R for a, b, c in zip(tdot(aL, wL), tdot(aR, wR), tdot(aO, wO))]
    return (wL, wR, wO, cs), (aL, aR, aO)


def selfcheck():
    """Numerical checks of the constants."""
    def is_probable_prime(n):
        import random
        rr = random.Random(1)
        if n % 2 == 0:
            return False
        d, s = n - 1, 0
        while d % 2 == 0:
            d //= 2
            s += 1
        for _ in range(24):
            a = rr.randrange(2, n - 1)
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(s - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True

    assert is_probable_prime(Q) and Q.bit_length() == 381
    assert is_probable_prime(R) and R.bit_length() == 255
    assert g1_is_on_curve(G1_GEN)
    assert g1_mul(G1_GEN, R - 1) == g1_neg(G1_GEN)
    assert g1_add(g1_mul(G1_GEN, R - 1), G1_GEN) is INF
    bx = -0xD201000000010000
    assert R == bx ** 4 - bx ** 2 + 1
    assert Q == (bx - 1) ** 2 * R // 3 + bx
    assert (R - 1) % (1 << 32) == 0 and (R - 1) % (1 << 33) != 0
    assert pow(7, (R - 1) // 2, R) == R - 1
    return True


if __name__ == "__main__":
    import random
    selfcheck()
    rng = random.Random(0)
    circ, asg = arith_circuit_example2(12)
    srs = SRS(50, rng.randrange(1, R), rng.randrange(1, R))
    tr = [rng.randrange(R) for _ in range(transcript_len(5))]
    proof, _ = prove(srs, asg, circ, tr)
    print("example2 verify:", verify_exponent(srs, circ, asg, tr, proof))
    print(proof_to_bytes(proof).hex()[:64], "...")
