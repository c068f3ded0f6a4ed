"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bit-exact (integer / byte work): every comparison is ==."""
import ctypes as C
import random

import numpy as np
import pytest

from util import NCPU, R, big_circuit, circuit_arrays, fr_bytes, rand_fr_array

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def srs_pair(sonic, orc):
    """One SRS on both sides (d = 2^12): GPU-generated, and the oracle's."""
    pyr = random.Random(11)
    d = 1 << 12
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    return d, x, alpha, sonic.SRS.new(d, x, alpha), orc.SRS(d, x, alpha, threads=NCPU)


def test_srs_new_matches_oracle(srs_pair):
    """SRS.new (SRS.hs:27-43): every G1 element of both bases, including the omitted g^alpha slot."""
    d, x, alpha, g, o = srs_pair
    for basis in (0, 1):
        assert np.array_equal(g.points(basis, -d, 2 * d + 1), o.points(basis, -d, 2 * d + 1))
    assert g.points(1, 0, 1).tobytes() == bytes(96)


def test_public_known_answer_vectors_on_gpu(sonic):
    """EIP-2537's 2 G1 and 2 G2 (tests/golden/eip2537_kat.json, public, independent of repository and reference) from the HIP
    path: SRS.new with x = 2 puts them at gPositiveX[1] / hPositiveX[1]; the MSM entry point gives 2 G1 as 2 * G and G + G"""
    import json
    import os
    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eip2537_kat.json")))
    g = (int(k["g1_generator"]["x"], 16), int(k["g1_generator"]["y"], 16))
    g2x = (int(k["g1_generator_doubled"]["x"], 16), int(k["g1_generator_doubled"]["y"], 16))
    h2 = k["g2_generator_doubled"]
    h2x = ((int(h2["x_c0"], 16), int(h2["x_c1"], 16)), (int(h2["y_c0"], 16), int(h2["y_c1"], 16)))
    s = sonic.SRS.new(8, 2, 3)
    assert s.gPositiveX(0) == g and s.gPositiveX(1) == g2x
    assert s.hPositiveX(1) == h2x
    pts = np.frombuffer(sonic.g1_to_bytes(g) * 2, np.uint8).reshape(2, 96)
    assert sonic.g1_from_bytes(sonic.msm_g1(pts[:1], [2])) == g2x
    assert sonic.g1_from_bytes(sonic.msm_g1(pts, [1, 1])) == g2x


def test_srs_index_maps(srs_pair, ref):
    """the four reference vectors as views (SRS.hs:33-39), against the literal python restatement"""
    d, x, alpha, g, _ = srs_pair
    s = ref.SRS(d, x, alpha)
    for k in (0, 1, 5, d - 1):
        assert g.gNegativeX(k) == s.gNegativeX(k)
        assert g.gPositiveX(k) == s.gPositiveX(k)
        assert g.gNegativeAlphaX(k) == s.gNegativeAlphaX(k)
        assert g.gPositiveAlphaX(k) == s.gPositiveAlphaX(k)
    assert g.gPositiveX(d) == s.gPositiveX(d)
    with pytest.raises(IndexError):
        g.gPositiveAlphaX(d)


@pytest.mark.parametrize("n,kind", [(0, "rand"), (1, "rand"), (2, "rand"), (63, "rand"), (64, "edge"), (65, "rand"),
                                    (1000, "rand"), (1024, "same"), (4097, "edge"), (8000, "same"), (8191, "rand")])
def test_msm_srs_slices(sonic, orc, srs_pair, n, kind):
    """the commitPoly/openPoly fold (CommitmentScheme.hs:26-29,45-48) on SRS slices; edge scalars
    0, 1, r-1, (r+-1)/2; n equal scalars (heavy-bucket path)."""
    from sonic_amd.commitment import msm_g1_srs
    d, _, _, g, o = srs_pair
    pyr = random.Random(n * 7 + len(kind))
    if kind == "rand":
        vals = [pyr.randrange(R) for _ in range(n)]
    elif kind == "edge":
        vals = [[0, 1, R - 1, 2, R - 2, (R - 1) // 2, (R + 1) // 2][i % 7] for i in range(n)]
    else:
        vals = [pyr.randrange(R)] * n
    sc = fr_bytes(vals)
    for basis in (0, 1):
        e0 = 1 if (basis == 1 and n <= d) else -(n // 2)
        assert msm_g1_srs(g, basis, e0, sc) == orc.msm_srs(o, basis, e0, sc, 1, NCPU)


def test_srs_file_round_trip(sonic, orc, srs_pair, tmp_path):
    """SRS on disk: save -> load gives the same points (validated on load), the same commitments, and -- with the G2 half in
    the file -- a handle that verifies without ever having seen x or alpha; corrupt files are refused"""
    d, x, alpha, g, o = srs_pair
    path = tmp_path / "srs.bin"
    g.save(path)
    n = 2 * d + 1
    assert path.stat().st_size == 24 + 2 * n * 96 + 2 * n * 192
    l = sonic.SRS.load(path)
    assert l.srsD == d
    for basis in (0, 1):
        assert np.array_equal(l.points(basis, -d, n), g.points(basis, -d, n))
        assert np.array_equal(l.g2_points(basis, -d, n), g.g2_points(basis, -d, n))
    f = {e: (e * e + 7) % R for e in range(-300, 200) if e != 0}
    F = sonic.commit_poly(l, d, f)
    assert F == sonic.commit_poly(g, d, f)
    op = sonic.open_poly(l, 12345, f)
    assert sonic.pc_v(l, d, F, 12345, op) and not sonic.pc_v(l, d, F, 12346, op)       # pairing check on the loaded handle
    g1only = tmp_path / "g1.bin"
    g.save(g1only, g2=False)
    assert g1only.stat().st_size == 24 + 2 * n * 96
    l1 = sonic.SRS.load(g1only)
    assert sonic.commit_poly(l1, d, f) == F
    with pytest.raises(sonic.SonicError):
        l1.hPositiveX(0)                             # G1-only file: no G2 half, and no trapdoor to make one from
    l1.set_g2_points(g.g2_points(0, -d, n), g.g2_points(1, -d, n))
    assert l1.hPositiveX(1) == g.hPositiveX(1) and sonic.pc_v(l1, d, F, 12345, op)
    raw = bytearray(path.read_bytes())
    raw[24 + 96 * 5] ^= 1                            # a G1 point off the curve
    bad = tmp_path / "bad.bin"
    bad.write_bytes(bytes(raw))
    with pytest.raises(sonic.SonicError) as e:
        sonic.SRS.load(bad)
    assert e.value.code == 3
    raw = bytearray(path.read_bytes())
    raw[24 + 2 * n * 96 + 192 * 7 + 3] ^= 1          # a G2 point off the twist
    bad.write_bytes(bytes(raw))
    with pytest.raises(sonic.SonicError) as e:
        sonic.SRS.load(bad)
    assert e.value.code == 3
    bad.write_bytes(bytes(raw[:1000]))
    with pytest.raises(sonic.SonicError):
        sonic.SRS.load(bad)
    bad.write_bytes(path.read_bytes() + b"\0")
    with pytest.raises(sonic.SonicError):
        sonic.SRS.load(bad)


def test_srs_refuses_points_at_infinity(sonic, srs_pair, tmp_path):
    """no power of a generator is the identity (x, alpha != 0): an SRS file whose G2 section is zero-filled must not load -- a
    verifier key at infinity pairs to 1 and would accept every proof -- and neither may a G1 basis with an empty slot other than
    the omitted g^alpha.  A handle without a G2 half saves (default: G2 only if present) and reloads as it is."""
    d, x, alpha, g, o = srs_pair
    n = 2 * d + 1
    path = tmp_path / "srs.bin"
    g.save(path)
    assert g.has_g2()
    raw = bytearray(path.read_bytes())
    g2_off = 24 + 2 * n * 96
    bad = tmp_path / "bad.bin"
    z = bytearray(raw)
    z[g2_off:] = bytes(len(raw) - g2_off)              # whole G2 section zeroed
    bad.write_bytes(bytes(z))
    with pytest.raises(sonic.SonicError) as e:
        sonic.SRS.load(bad)
    assert e.value.code == 3 and "infinity" in e.value.message
    z = bytearray(raw)
    k = g2_off + 192 * (n + d + 1)                     # one element: hPositiveAlphaX[1], a verifier-key element
    z[k:k + 192] = bytes(192)
    bad.write_bytes(bytes(z))
    with pytest.raises(sonic.SonicError) as e:
        sonic.SRS.load(bad)
    assert e.value.code == 3
    z = bytearray(raw)
    z[24 + 96 * 3:24 + 96 * 4] = bytes(96)             # a G1 element of basis 0
    bad.write_bytes(bytes(z))
    with pytest.raises(sonic.SonicError) as e:
        sonic.SRS.load(bad)
    assert e.value.code == 3 and "infinity" in e.value.message
    b0, b1 = g.points(0, -d, n), g.points(1, -d, n)
    assert not b1[d].any()                             # the omitted g^alpha (SRS.hs:38) is the one empty slot
    with pytest.raises(sonic.SonicError):
        sonic.SRS.from_points(d, np.zeros_like(b0), b1)
    h = sonic.SRS.from_points(d, b0, b1)
    assert not h.has_g2()
    with pytest.raises(sonic.SonicError):
        h.set_g2_points(np.zeros((n, 192), np.uint8), np.zeros((n, 192), np.uint8))
    assert not h.has_g2()
    p1 = tmp_path / "g1only.bin"
    h.save(p1)                                         # default: the G2 half only if the handle has it
    assert p1.stat().st_size == 24 + 2 * n * 96
    with pytest.raises(sonic.SonicError) as e:
        h.save(p1, g2=True)
    assert e.value.code == 7
    l = sonic.SRS.load(p1)
    assert not l.has_g2() and np.array_equal(l.points(1, -d, n), b1)
    l.save(tmp_path / "again.bin")
    assert (tmp_path / "again.bin").read_bytes() == p1.read_bytes()


def test_msm_entry_encoding_limits(sonic, srs_pair):
    """a sorted entry packs the term index into 26 bits (+ 5 window bits) over window tables and into 31 bits otherwise: an MSM
    of 2^26 terms and more is planned over per-window buckets, one of 2^31 terms is refused -- by check, not by running out of
    memory first"""
    import ctypes as C
    from sonic_amd import _lib
    L = _lib.lib()
    d, x, alpha, g, o = srs_pair
    c, w, sets = C.c_int(), C.c_int(), C.c_int()
    _lib.check(L.sonic_msm_plan(g._h, (1 << 26) - 1, C.byref(c), C.byref(w), C.byref(sets)))
    tables = sets.value == 1
    assert w.value <= 32 or not tables
    _lib.check(L.sonic_msm_plan(g._h, 1 << 26, C.byref(c), C.byref(w), C.byref(sets)))
    assert sets.value == w.value and w.value <= 64      # no shared bucket set: the index field of the table path would overflow
    assert L.sonic_msm_plan(g._h, 1 << 31, C.byref(c), C.byref(w), C.byref(sets)) == 7
    assert "2^31" in _lib.last_error()


def test_srs_points_must_be_in_the_subgroup(sonic, srs_pair):
    """E(Fq) has cofactor points -- (0, 2) has order 3 -- and MSMs over an SRS fold scalars with r P = O, so the record
    constructor refuses points outside the order-r subgroup; the same for the G2 vectors (a point on the twist is almost
    never in G2: the cofactor is ~2^508)"""
    d, x, alpha, g, o = srs_pair
    n = 2 * d + 1
    b0, b1 = g.points(0, -d, n), g.points(1, -d, n)
    ok = sonic.SRS.from_points(d, b0, b1)
    assert np.array_equal(ok.points(0, -d, n), b0)
    bad = b0.copy()
    bad[17] = np.frombuffer((0).to_bytes(48, "little") + (2).to_bytes(48, "little"), np.uint8)
    with pytest.raises(sonic.SonicError) as e:
        sonic.SRS.from_points(d, bad, b1)
    assert e.value.code == 3 and "subgroup" in e.value.message
    # a point on the twist y^2 = x^3 + 4(u+1) outside G2: scan x = k (in Fq) until x^3 + 4(u+1) is a square in Fq2
    from oracle import pairing as pg
    Qm = pg.Q

    def fq_sqrt(a):                                  # q = 3 mod 4
        s_ = pow(a, (Qm + 1) // 4, Qm)
        return s_ if s_ * s_ % Qm == a % Qm else None

    def f2_sqrt(a):
        s_ = fq_sqrt((a[0] * a[0] + a[1] * a[1]) % Qm)
        if s_ is None:
            return None
        inv2 = pow(2, -1, Qm)
        for t in ((a[0] + s_) * inv2 % Qm, (a[0] - s_) * inv2 % Qm):
            x0 = fq_sqrt(t)
            if x0:
                y_ = (x0, a[1] * pow(2 * x0, -1, Qm) % Qm)
                if pg.f2_sqr(y_) == (a[0] % Qm, a[1] % Qm):
                    return y_
        return None

    pt = None
    for k in range(1, 200):
        y = f2_sqrt(pg.f2_add(pg.f2_mul(pg.f2_sqr((k, 0)), (k, 0)), (4, 4)))
        if y is not None:
            pt = ((k, 0), y)
            break
    assert pt is not None
    h0, h1 = g.g2_points(0, -d, n), g.g2_points(1, -d, n)
    enc = b"".join(int(v).to_bytes(48, "little") for v in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]))
    h0b = h0.copy()
    h0b[3] = np.frombuffer(enc, np.uint8)
    with pytest.raises(sonic.SonicError) as e:
        ok.set_g2_points(h0b, h1)
    assert e.value.code == 3 and "subgroup" in e.value.message
    h0b[3, 5] ^= 1
    with pytest.raises(sonic.SonicError) as e:
        ok.set_g2_points(h0b, h1)
    assert e.value.code == 3 and "twist" in e.value.message
    with pytest.raises(sonic.SonicError):
        ok.hPositiveX(0)                             # nothing was attached by the refused calls
    ok.set_g2_points(h0, h1)
    assert ok.hPositiveAlphaX(1) == g.hPositiveAlphaX(1)


def test_srs_g2_half(sonic, srs_pair):
    """the G2 vectors of SRS.new (SRS.hs:35-36,40-41) against python big-integer G2 arithmetic (oracle/pairing.py)"""
    from oracle import pairing as pg
    d, x, alpha, g, _ = srs_pair
    s = pg.SRS(d, x, alpha)
    for k in (0, 1, 7, d - 1):
        assert g.hNegativeX(k) == s.hNegativeX(k)
        assert g.hPositiveX(k) == s.hPositiveX(k)
        assert g.hPositiveAlphaX(k) == s.hPositiveAlphaX(k)
        assert g.hNegativeAlphaX(k) == pg.g2_mul(pg.G2_GEN, alpha * pow(s.x_inv, k + 1, R) % R)
    assert g.hPositiveX(d) == s.hPositiveX(d) and g.hPositiveX(0) == pg.G2_GEN
    with pytest.raises(IndexError):
        g.hNegativeX(d)
    blk = g.g2_points(0, -5, 11)
    for i in (0, 4, 5, 10):
        p_ = s.hPositiveX(i - 5) if i >= 5 else s.hNegativeX(4 - i)
        assert blk[i].tobytes() == b"".join(v.to_bytes(48, "little") for v in (p_[0][0], p_[0][1], p_[1][0], p_[1][1]))


def test_window_tables(sonic, orc, srs_pair):
    """the precomputed window tables (table w = 2^shift(w) * basis, even window widths) behind the shared-bucket MSM, read back through the
    diagnostic basis index b + 2w; and an SRS built with the tables switched off gives the same MSM"""
    import os
    from sonic_amd.commitment import msm_g1_srs
    d, x, alpha, g, o = srs_pair
    import ctypes as C
    from sonic_amd import _lib
    pc, pw, pb = C.c_int(), C.c_int(), C.c_int()
    _lib.check(_lib.lib().sonic_msm_plan(g._h, 5000, C.byref(pc), C.byref(pw), C.byref(pb)))
    c, W = pc.value, pw.value
    base, extra = 255 // W, 255 % W                                                  # even window widths (msm.hpp): 255 % W wide ones first
    assert pb.value == 1 and c == base + (1 if extra else 0)                       # shared buckets; c is the widest window
    shift = lambda w: w * base + min(w, extra)
    P = o.points(1, -d, 2 * d + 1)
    for w in (1, 2, extra, W - 1):
        T = g.points(1 + 2 * w, -d, 2 * d + 1)
        for i in (0, 1, 63, 64, 65, 4095, 4096, 4097, 8192):
            assert T[i].tobytes() == orc.g1_mul(P[i].tobytes(), pow(2, shift(w), R)), (w, i)
    os.environ["SONIC_MSM_TABLES"] = "0"
    try:
        plain = sonic.SRS.new(d, x, alpha)
    finally:
        del os.environ["SONIC_MSM_TABLES"]
    with pytest.raises(sonic.SonicError):
        plain.points(2, 0, 1)                    # no table 1
    sc = rand_fr_array(np.random.default_rng(77), 5000)
    assert msm_g1_srs(plain, 1, -2500, sc) == msm_g1_srs(g, 1, -2500, sc) == orc.msm_srs(o, 1, -2500, sc, 1, NCPU)
    # SONIC_MSM_TABLE_C: another window width for the tables (the A/B knob behind profiles/r05_table_c_ab.txt): other windows, same sums
    os.environ["SONIC_MSM_TABLE_C"] = "9"
    try:
        narrow = sonic.SRS.new(d, x, alpha)
    finally:
        del os.environ["SONIC_MSM_TABLE_C"]
    _lib.check(_lib.lib().sonic_msm_plan(narrow._h, 5000, C.byref(pc), C.byref(pw), C.byref(pb)))
    assert (pc.value, pw.value, pb.value) == (9, 29, 1) and (c, W) != (9, 29)
    assert msm_g1_srs(narrow, 1, -2500, sc) == msm_g1_srs(g, 1, -2500, sc)


@pytest.mark.parametrize("c", [4, 7, 11, 16])
def test_msm_window_sizes(sonic, orc, srs_pair, c):
    from sonic_amd import _lib
    from sonic_amd.commitment import msm_g1_srs
    d, _, _, g, o = srs_pair
    sc = rand_fr_array(np.random.default_rng(c), 3000)
    _lib.lib().sonic_msm_set_window(c)
    try:
        assert msm_g1_srs(g, 0, -1500, sc) == orc.msm_srs(o, 0, -1500, sc, 1, NCPU)
    finally:
        _lib.lib().sonic_msm_set_window(0)


def test_msm_general_points(sonic, orc, srs_pair):
    """caller-supplied points: repeated points (P+P), P and -P, infinity, x = 1 style all-equal bases
    (bench/Main.hs:23 makes every SRS element equal g)"""
    d, _, _, g, o = srs_pair
    n = 500
    pts = o.points(0, -10, n).copy()
    pts[10] = pts[11]
    pts[20] = 0
    neg = pts[30].copy()
    y = int.from_bytes(neg[48:].tobytes(), "little")
    Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    neg[48:] = np.frombuffer((Q - y).to_bytes(48, "little"), np.uint8)
    pts[31] = neg
    sc = rand_fr_array(np.random.default_rng(5), n)
    sc[31] = sc[30]          # s*P + s*(-P) = O
    assert sonic.msm_g1(pts, sc) == orc.msm(pts, sc, 1, NCPU)
    same = np.repeat(pts[:1], 300, axis=0)
    sc2 = rand_fr_array(np.random.default_rng(6), 300)
    assert sonic.msm_g1(same, sc2) == orc.msm(same, sc2, 1, NCPU)


def test_msm_rejects_bad_encoding(sonic, srs_pair):
    d, _, _, g, o = srs_pair
    pts = o.points(0, 0, 4).copy()
    sc = fr_bytes([1, 2, 3, 4])
    bad = sc.copy(); bad[2] = 0xFF                     # >= r
    with pytest.raises(sonic.SonicError) as e:
        sonic.msm_g1(pts, bad)
    assert e.value.code == 3
    off = pts.copy(); off[1, 0] ^= 1                   # not on the curve
    with pytest.raises(sonic.SonicError) as e:
        sonic.msm_g1(off, sc)
    assert e.value.code == 3
    from sonic_amd.encoding import Q_MODULUS
    qq = pts.copy(); qq[2] = np.frombuffer(Q_MODULUS.to_bytes(48, "little") * 2, np.uint8)   # (q, q): congruent to (0, 0) but not O
    with pytest.raises(sonic.SonicError) as e:
        sonic.msm_g1(qq, sc)
    assert e.value.code == 3
    xq = pts.copy()                                    # x + q: the same residue, non-canonical bytes
    xv = int.from_bytes(xq[3, :48].tobytes(), "little") + Q_MODULUS
    if xv < 1 << 384:
        xq[3, :48] = np.frombuffer(xv.to_bytes(48, "little"), np.uint8)
        with pytest.raises(sonic.SonicError) as e:
            sonic.msm_g1(xq, sc)
        assert e.value.code == 3


def test_msm_point_with_zero_coordinate(sonic, orc, srs_pair):
    """(0, 2) is on y^2 = x^3 + 4 (not in the r-torsion, but the group law does not care): a coordinate congruent to 0 must
    not be taken for the point at infinity, whichever representative of 0 the device arithmetic produces"""
    d, _, _, g, o = srs_pair
    pts = o.points(0, 0, 6).copy()
    z2 = np.frombuffer((0).to_bytes(48, "little") + (2).to_bytes(48, "little"), np.uint8)
    from sonic_amd.encoding import Q_MODULUS
    z2n = np.frombuffer((0).to_bytes(48, "little") + (Q_MODULUS - 2).to_bytes(48, "little"), np.uint8)
    pts[1] = z2; pts[4] = z2n
    for seed in range(3):
        sc = rand_fr_array(np.random.default_rng(40 + seed), 6)
        assert sonic.msm_g1(pts, sc) == orc.msm(pts, sc, 1, NCPU)
    sc = fr_bytes([0, 5, 0, 0, 5, 0])                  # 5*(0,2) + 5*(0,-2) = O
    assert sonic.msm_g1(pts, sc) == bytes(96) == orc.msm(pts, sc, 1, NCPU)
    sc = fr_bytes([0, 3, 0, 0, 0, 0])                  # 3*(0,2): (0,2) has order 3, so this is O as well
    assert sonic.msm_g1(pts, sc) == orc.msm(pts, sc, 1, NCPU)


@pytest.mark.parametrize("log2n", [0, 1, 3, 10, 11, 12, 13, 14, 15, 17, 18])      # 0, 1, 2, 3, 4, 6, 7 wide stages (radix-4 passes + an odd radix-2 one)
def test_ntt_matches_oracle(sonic, orc, log2n):
    from sonic_amd import _lib
    n = 1 << log2n
    a = rand_fr_array(np.random.default_rng(log2n), n)
    for inverse in (0, 1):
        got = a.copy()
        _lib.check(_lib.lib().sonic_ntt_fr(got.ctypes.data, log2n, inverse))
        assert np.array_equal(got, orc.ntt(a, bool(inverse)))
    rt = a.copy()
    _lib.check(_lib.lib().sonic_ntt_fr(rt.ctypes.data, log2n, 0))
    _lib.check(_lib.lib().sonic_ntt_fr(rt.ctypes.data, log2n, 1))
    assert np.array_equal(rt, a)


def test_ntt_size_limit(sonic):
    """2^28 points is the largest transform (32-bit unsigned twiddle offsets inside the butterfly routines; tests/test_gpu_fullsize.py runs
    one): beyond it the call is refused before anything is read"""
    from sonic_amd import _lib
    buf = np.zeros((4, 32), np.uint8)
    for log2n in (29, 40, -1):
        assert _lib.lib().sonic_ntt_fr(buf.ctypes.data, log2n, 0) == 7        # SONIC_ERR_INVALID_ARG


@pytest.mark.parametrize("log2n", [11, 12, 16])
def test_ntt_extreme_inputs(sonic, orc, log2n):
    """the assembly butterflies keep values in [0, 2r) between stages (sums pass 2^256, differences borrow): inputs that sit on the
    edges of the field -- all r - 1, zeros and r - 1 alternating, blocks of the two, a single non-zero element, a constant -- must
    still give the oracle's canonical transform in both directions"""
    from sonic_amd import _lib
    n = 1 << log2n
    top = np.frombuffer((R - 1).to_bytes(32, "little"), np.uint8)
    one = np.frombuffer((1).to_bytes(32, "little"), np.uint8)
    cases = []
    a = np.tile(top, (n, 1)); cases.append(a)
    a = np.zeros((n, 32), np.uint8); a[::2] = top; cases.append(a)
    a = np.zeros((n, 32), np.uint8); a[1::2] = top; a[::2] = one; cases.append(a)
    a = np.zeros((n, 32), np.uint8); a[: n // 2] = top; cases.append(a)
    a = np.zeros((n, 32), np.uint8); a[n - 1] = top; cases.append(a)
    a = np.zeros((n, 32), np.uint8); a[(np.arange(n) // 1024) % 2 == 1] = top; cases.append(a)
    for a in cases:
        a = np.ascontiguousarray(a)
        for inverse in (0, 1):
            got = a.copy()
            _lib.check(_lib.lib().sonic_ntt_fr(got.ctypes.data, log2n, inverse))
            assert np.array_equal(got, orc.ntt(a, bool(inverse)))


@pytest.mark.parametrize("na,nb", [(1, 1), (3, 5), (100, 37), (1025, 1024), (5000, 7000)])
def test_poly_mul_matches_schoolbook(sonic, orc, na, nb):
    """the `*` of Constraints.hs:61 as a dense product, against the oracle's schoolbook convolution"""
    from sonic_amd import _lib
    g = np.random.default_rng(na * 31 + nb)
    a, b = rand_fr_array(g, na), rand_fr_array(g, nb)
    out = np.zeros((na + nb - 1, 32), np.uint8)
    _lib.check(_lib.lib().sonic_poly_mul_fr(a.ctypes.data, na, b.ctypes.data, nb, out.ctypes.data))
    assert np.array_equal(out, orc.poly_mul(a, b, use_ntt=(na * nb > 4_000_000)))


def _sparse_poly(pyr, lo, hi, density=0.7, hole=None):
    exps = [e for e in range(lo, hi + 1) if pyr.random() < density and e != hole]
    return {e: pyr.randrange(1, R) for e in exps}


def test_commit_open_match_oracle(sonic, orc, srs_pair):
    """commitPoly / openPoly on sparse Laurent polynomials: both bases, shifts (max = n and max = d),
    duplicates summed, zero coefficients ignored"""
    d, x, alpha, g, o = srs_pair
    pyr = random.Random(3)
    for (lo, hi, maxm) in [(-40, 30, d), (-200, 100, 100), (-1, 1, d), (5, 9, d), (-9, -5, d), (-3000, 2500, d)]:
        hole = maxm - d if lo <= maxm - d <= hi else None
        f = _sparse_poly(pyr, lo, hi, hole=0 if maxm == d else hole)
        exps = np.array(sorted(f), np.int64)
        co = fr_bytes([f[e] for e in sorted(f)])
        assert sonic.g1_to_bytes(sonic.commit_poly(g, maxm, f)) == orc.commit_poly(o, maxm, exps, co)
        z = pyr.randrange(1, R)
        fz, W = sonic.open_poly(g, z, f)
        ofz, oW = orc.open_poly(o, z, exps, co)
        assert fz == ofz and sonic.g1_to_bytes(W) == oW
    # repeated exponents are summed, explicit zeros are harmless
    f = [(3, 5), (3, R - 5), (2, 7), (-4, 0), (1, 9)]
    assert sonic.commit_poly(g, d, f) == sonic.commit_poly(g, d, {2: 7, 1: 9})


def test_commit_error_contract(sonic, srs_pair):
    """`index` panics (CommitmentScheme.hs:70-73): exponent past the vector end, and the e' = 0 hole"""
    d, _, _, g, _ = srs_pair
    with pytest.raises(sonic.SonicError) as e:
        sonic.commit_poly(g, d, {0: 5, 1: 1})
    assert e.value.code == 2
    with pytest.raises(sonic.SonicError) as e:
        sonic.commit_poly(g, d, {d + 1: 5})
    assert e.value.code == 2
    with pytest.raises(sonic.SonicError) as e:
        sonic.open_poly(g, 7, {d + 2: 5, 0: 1})
    assert e.value.code == 2
    with pytest.raises(sonic.SonicError) as e:
        sonic.open_poly(g, 0, {-2: 5, 1: 1})
    assert e.value.code == 4


@pytest.mark.parametrize("n,Q", [(1, 1), (2, 2), (3, 1), (8, 2), (20, 5), (64, 3), (257, 2)])
def test_prove_matches_oracle(sonic, orc, ref, srs_pair, n, Q):
    """prove (Protocol.hs:47-109 + Signature.hs:38-72): proof bytes equal the oracle's on the same
    circuit, SRS and transcript; rndCircuit generator of test/Test/Reference.hs:125-169"""
    d, x, alpha, g, o = srs_pair
    pyr = random.Random(n * 100 + Q)
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
    want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(tr))
    circuit = sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3])
    proof, oracle = sonic.prove(g, sonic.Assignment(*asg), circuit, transcript=tr)
    assert proof.to_bytes() == want
    assert oracle.rndOracleY == tr[4] and oracle.rndOracleZ == tr[5]
    assert proof.prHscProof.hscU == tr[6 + 2 * Q] and proof.prHscProof.hscV == tr[7 + 2 * Q]


def test_prove_example_circuits(sonic, ref, srs_pair):
    """arithCircuitExample1/2 (test/Test/Reference.hs:38-90, examples/Main.hs:38-63) against the literal
    python restatement, plus the reference's own acceptance test (verify . prove) through the trapdoor"""
    d, x, alpha, g, _ = srs_pair
    s = ref.SRS(d, x, alpha)
    pyr = random.Random(99)
    for circ, asg in (ref.arith_circuit_example1(), ref.arith_circuit_example2(12)):
        Q = len(circ[0])
        tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
        proof, _ = sonic.prove(g, sonic.Assignment(*asg), sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), circ[3]), transcript=tr)
        want, _ = ref.prove(s, asg, circ, tr)
        assert proof.to_bytes() == ref.proof_to_bytes(want)
        assert ref.verify_exponent(s, circ, asg, tr, want)


def test_reference_bench_shape(sonic, orc, ref):
    """the reference's criterion benchmark (bench/Main.hs:18-27,37-49): x = 1 (!), alpha = 4, d = 25 n, Example1 and Example2.
    With x = 1 every element of a basis is the SAME point, so every bucket walk is a chain of P + P / P + (-P): the exceptional
    lanes of the fused additions (both variants) and of the running sums carry the whole computation.  Prover bytes against
    both oracles, and the verifier accepts; a larger random circuit on the same degenerate SRS for the sorted-bucket paths."""
    pyr = random.Random(2024)
    for circ, asg in (ref.arith_circuit_example1(), ref.arith_circuit_example2(12)):
        n, Q = len(asg[0]), len(circ[0])
        d = 25 * n
        g, s = sonic.SRS.new(d, 1, 4), ref.SRS(d, 1, 4)
        tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
        circuit = sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), circ[3])
        proof, ro = sonic.prove(g, sonic.Assignment(*asg), circuit, transcript=tr)
        want, _ = ref.prove(s, asg, circ, tr)
        assert proof.to_bytes() == ref.proof_to_bytes(want)
        assert sonic.verify(g, circuit, proof, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
    n, Q, d = 300, 3, 7 * 300 + 5
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
    g, o = sonic.SRS.new(d, 1, 4), orc.SRS(d, 1, 4, threads=NCPU)
    p = sonic.Prover(g, sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3]))
    p.set_assignment(sonic.Assignment(*asg))
    assert p.prove_bytes(tr) == orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(tr))


def test_prove_error_contract(sonic, ref, srs_pair):
    d, x, alpha, g, _ = srs_pair
    pyr = random.Random(5)
    n = d // 7 + 1                                      # d < 7n  -> Protocol.hs:54-55
    circ, asg = ref.rnd_circuit(pyr, n, 1)
    with pytest.raises(sonic.SonicError) as e:
        sonic.prove(g, sonic.Assignment(*asg), sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), circ[3]))
    assert e.value.code == 1
    # unsatisfied constraint system: t(X,y) has a constant term -> commitPoly indexes -1
    circ, asg = ref.rnd_circuit(pyr, 4, 2)
    bad_cs = list(circ[3]); bad_cs[0] = (bad_cs[0] + 1) % R
    with pytest.raises(sonic.SonicError) as e:
        sonic.prove(g, sonic.Assignment(*asg), sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), bad_cs))
    assert e.value.code == 2


def test_one_shot_calls_into_a_parked_shell(sonic, orc, ref, srs_pair):
    """sonic_prove (Protocol.hs:47-52: everything per call) re-uses the shell of the previous call with the same SRS and (n, Q) and uploads
    the new circuit INSIDE the proof, under the MSMs that need the assignment only: different circuits through one shell give the oracle's
    bytes each; a non-canonical weight is reported with the status the circuit upload always had (and no proof), and the shell still
    serves the next call"""
    import ctypes as C
    from sonic_amd import _lib
    d, x, alpha, g, o = srs_pair
    n, Q = 150, 3
    pyr = random.Random(77)
    L = _lib.lib()
    out = (C.c_uint8 * L.sonic_proof_size(Q))()
    for trial in range(4):
        circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
        tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
        args = [np.ascontiguousarray(enc[k]) for k in ("wL", "wR", "wO", "cs", "aL", "aR", "aO")]
        if trial == 2:
            args[1] = args[1].copy()
            args[1].reshape(-1, 32)[n + 5] = np.frombuffer(R.to_bytes(32, "little"), np.uint8)        # wR[1][5] = r: not < r
        C.memset(out, 0, len(out))
        rc = L.sonic_prove(g._h, n, Q, *[a.ctypes.data for a in args], tr.ctypes.data, out)
        if trial == 2:
            assert rc == 3 and bytes(out) == bytes(len(out))                 # SONIC_ERR_BAD_ENCODING
            msg = C.create_string_buffer(512); L.sonic_last_error(msg, 512)
            assert b"non-canonical" in msg.value
        else:
            assert rc == 0
            assert bytes(out) == orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], tr)


def test_prover_handle_reuse(sonic, orc, ref, srs_pair):
    """circuit resident in HBM, several assignments / transcripts through one handle"""
    d, x, alpha, g, o = srs_pair
    pyr = random.Random(8)
    n, Q = 33, 2
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    p = sonic.Prover(g, sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3]))
    p.set_assignment(sonic.Assignment(*asg))
    for _ in range(3):
        tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
        want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], tr)
        assert p.prove_bytes(tr) == want


def test_submit_collect_pipeline(sonic, orc, ref, srs_pair):
    """prove = submit + collect; two handles driven in turn by one host thread give the same bytes as proving one after the
    other (and as the oracle); the halves refuse to be misused"""
    d, x, alpha, g, o = srs_pair
    pyr = random.Random(77)
    n, Q = 40, 3
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    circuit = sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3])
    trs = [[pyr.randrange(1, R) for _ in range(8 + 2 * Q)] for _ in range(5)]
    want = [orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(t)) for t in trs]
    pipe = sonic.ProverPipeline(g, circuit, depth=2)
    pipe.set_assignment(sonic.Assignment(*asg))
    assert pipe.prove_all(trs) == want
    assert pipe.prove_all(trs[:1]) == want[:1] and pipe.prove_all([]) == []
    p = pipe.provers[0]
    with pytest.raises(sonic.SonicError) as e:
        p.collect()                                  # nothing submitted
    assert e.value.code == 7
    p.submit(trs[2])
    for misuse in (lambda: p.submit(trs[3]), lambda: p.prove_bytes(trs[3]), lambda: p.set_assignment(sonic.Assignment(*asg))):
        with pytest.raises(sonic.SonicError) as e:
            misuse()                                 # one proof in flight per handle
        assert e.value.code == 7
    assert p.collect() == want[2]
    assert p.prove_bytes(trs[4]) == want[4]          # and the handle is as good as new
    pipe.close()


def test_prepared_handle_same_bytes(sonic, orc, ref, srs_pair):
    """sonic_prover_prepare (S_j assembled from the per-constraint commitments) must not change a single byte:
    rndCircuit weights, dense random weights, and a constraint row that is all zero (its commitment is O)"""
    d, x, alpha, g, o = srs_pair
    pyr = random.Random(81)
    for n, Q, kind in ((33, 2, "rnd"), (40, 3, "dense"), (17, 3, "zero-row"), (1, 1, "rnd")):
        circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
        if kind != "rnd":
            wL = [[pyr.randrange(R) for _ in range(n)] for _ in range(Q)]
            wR = [[pyr.randrange(R) for _ in range(n)] for _ in range(Q)]
            wO = [[pyr.randrange(R) for _ in range(n)] for _ in range(Q)]
            if kind == "zero-row":
                wL[1] = [0] * n; wR[1] = [0] * n; wO[1] = [0] * n
            aL, aR, aO = asg
            cs = [(sum(a * b for a, b in zip(wL[q], aL)) + sum(a * b for a, b in zip(wR[q], aR)) +
                   sum(a * b for a, b in zip(wO[q], aO))) % R for q in range(Q)]
            circ = (wL, wR, wO, cs)
            enc = dict(enc, wL=fr_bytes([v for row in wL for v in row]), wR=fr_bytes([v for row in wR for v in row]),
                       wO=fr_bytes([v for row in wO for v in row]), cs=fr_bytes(cs))
        ac = sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3])
        plain, prep = sonic.Prover(g, ac, prepare=False), sonic.Prover(g, ac, prepare=True)
        plain.set_assignment(sonic.Assignment(*asg)); prep.set_assignment(sonic.Assignment(*asg))
        for _ in range(2):
            tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
            want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], tr)
            assert plain.prove_bytes(tr) == want, (n, Q, kind)
            assert prep.prove_bytes(tr) == want, (n, Q, kind)


@pytest.mark.parametrize("n,Q", [(9, 5), (12, 7), (10, 9), (300, 8)])
def test_many_constraints_group_paths(sonic, orc, ref, srs_pair, n, Q):
    """Q > 4: the per-constraint commitments of a prepared handle are combined by a Q-term MSM on the GPU instead of on the host;
    Q + 2 > 8: the s(u,Y) group (C, Q_1..Q_Q, Q_v) no longer fits one batched MSM chain and is split.  Prepared and plain
    handles must both match the oracle."""
    d, x, alpha, g, o = srs_pair
    pyr = random.Random(900 + 17 * n + Q)
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    ac = sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3])
    tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
    want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], tr)
    for prepare in (False, True):
        p = sonic.Prover(g, ac, prepare=prepare)
        p.set_assignment(sonic.Assignment(*asg))
        assert p.prove_bytes(tr) == want, (n, Q, prepare)
        assert p.prove_bytes(tr) == want, (n, Q, prepare, "repeat")
        p.close()


def test_prove_without_window_tables(sonic, orc, ref, srs_pair):
    """an SRS built with SONIC_MSM_TABLES=0 has no shared-bucket plan: the MSM groups of prove() then run one MSM after the
    other over per-window buckets, and must give the same proof"""
    import os
    d, x, alpha, g, o = srs_pair
    os.environ["SONIC_MSM_TABLES"] = "0"
    try:
        plain = sonic.SRS.new(d, x, alpha)
    finally:
        del os.environ["SONIC_MSM_TABLES"]
    pyr = random.Random(4242)
    n, Q = 50, 3
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    ac = sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3])
    tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
    want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], tr)
    for prepare in (False, True):
        p = sonic.Prover(plain, ac, prepare=prepare)
        p.set_assignment(sonic.Assignment(*asg))
        assert p.prove_bytes(tr) == want, prepare
        p.close()


def test_prove_graph_replay(sonic, orc, ref, srs_pair):
    """SONIC_PROVE_GRAPH=1: the second proof of a handle is captured as a hipGraph (multi-stream capture) and later proofs replay
    it; every proof must still match the oracle, with new transcripts and a new assignment"""
    import os
    d, x, alpha, g, o = srs_pair
    pyr = random.Random(515)
    n, Q = 40, 2
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    os.environ["SONIC_PROVE_GRAPH"] = "1"
    try:
        p = sonic.Prover(g, sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3]))
    finally:
        del os.environ["SONIC_PROVE_GRAPH"]
    p.set_assignment(sonic.Assignment(*asg))
    for k in range(5):
        if k == 3:      # same circuit, another satisfying assignment is not available cheaply: re-upload the same one
            p.set_assignment(sonic.Assignment(*asg))
        tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
        want = orc.prove(o, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], tr)
        assert p.prove_bytes(tr) == want, k
    p.close()


def test_reference_verifier_accepts_gpu_proofs(sonic, ref):
    """the reference's only end-to-end test, verify . prove (test/Test/Protocol.hs:14-23), with the proof made by
    the HIP path and the verifier restated with real pairings (oracle/pairing.py: pcV, hscVerify, verify)"""
    from oracle import pairing as pg
    pyr = random.Random(31)
    for n, Q in ((1, 1), (3, 2)):
        circ, asg = ref.rnd_circuit(pyr, n, Q)
        d = {1: 12}.get(n, 7 * n) + pyr.randrange(5)
        x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
        g = sonic.SRS.new(d, x, alpha)
        proof, ro = sonic.prove(g, sonic.Assignment(*asg), sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), circ[3]), rng=pyr)
        class GpuG2(pg.SRS):             # the verifier reads its three G2 elements from the GPU-generated SRS
            def hNegativeX(self, k): return g.hNegativeX(k)
            def hPositiveX(self, k): return g.hPositiveX(k)
            def hPositiveAlphaX(self, k): return g.hPositiveAlphaX(k)
        vsrs = GpuG2(d, x, alpha)
        pr = pg.proof_from_bytes(proof.to_bytes(), Q)
        assert pg.verify(vsrs, circ, pr, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
        pr["prB"] = (pr["prB"] + 1) % R
        assert not pg.verify(vsrs, circ, pr, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)


def test_reference_example_program(sonic):
    """examples/Main.hs of the reference as examples/main.py: fresh x, alpha, z; SRS.new, prove, verify -> Success: True;
    and a wrong assignment does not verify (the prover refuses it: t(X,y) gets a constant term)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("sonic_example", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "main.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    assert ex.run_example() is True
    circuit, asg = ex.arith_circuit_example(5)
    assert ex.sonic_protocol_fs(circuit, asg, 12345) is True           # the same with the Fiat-Shamir transcript
    bad = sonic.Assignment(asg.aL, asg.aR, [(asg.aO[0] + 1) % R, asg.aO[1]])
    with pytest.raises(sonic.SonicError) as e:
        ex.sonic_protocol(circuit, bad, 12345)
    assert e.value.code == 2


def test_hsc_prove_and_verify_standalone(sonic, ref):
    """Sonic.Signature on its own (test/Test/Signature.hs:20-36): hscProve for the s(X,Y) of a random circuit with m (y_j, z_j)
    pairs, m independent of the number of constraints; every element against the literal restatement, hscVerify accepts it
    (host pairings, and the oracle's python pairing on the GPU's points), rejects a tampered one and other evaluation points"""
    from oracle import pairing as pg
    pyr = random.Random(55)
    for n, Q, m in ((3, 2, 3), (5, 4, 1), (2, 1, 0)):
        circ, asg = ref.rnd_circuit(pyr, n, Q)
        d = 7 * n + 3
        x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
        g, s = sonic.SRS.new(d, x, alpha), ref.SRS(d, x, alpha)
        circuit = sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), circ[3])
        yzs = [(pyr.randrange(1, R), pyr.randrange(1, R)) for _ in range(m)]
        u, v = pyr.randrange(1, R), pyr.randrange(1, R)
        got = sonic.hsc_prove(g, circuit, yzs, u, v)
        want = ref.hsc_prove(s, ref.s_poly(*circ[:3]), yzs, u, v)
        assert (got.hscS, got.hscW, got.hscQv, got.hscC, got.hscU, got.hscV) == \
            (want["hscS"], want["hscW"], want["hscQv"], want["hscC"], want["hscU"], want["hscV"])
        assert sonic.hsc_verify(g, circuit, yzs, got)
        if m:
            import dataclasses
            assert pg.hsc_verify(pg.SRS(d, x, alpha), ref.s_poly(*circ[:3]), yzs, want)
            sjp, wjp, qj = got.hscW[0]
            bad = dataclasses.replace(got, hscW=[((sjp + 1) % R, wjp, qj)] + got.hscW[1:])
            assert not sonic.hsc_verify(g, circuit, yzs, bad)
            assert not sonic.hsc_verify(g, circuit, [((yzs[0][0] + 1) % R, yzs[0][1])] + yzs[1:], got)
            assert not sonic.hsc_verify(g, circuit, yzs[:-1], got)
    fresh = sonic.hsc_prove(g, circuit, [(3, 5)])                      # u, v drawn inside, as the reference's `rnd`
    assert sonic.hsc_verify(g, circuit, [(3, 5)], fresh) and 0 < fresh.hscU < R and 0 < fresh.hscV < R
    with pytest.raises(sonic.SonicError) as e:
        sonic.hsc_prove(g, circuit, [(0, 5)], 1, 2)
    assert e.value.code == 4


def test_hsc_prove_arbitrary_bivariate_polynomial(sonic, ref):
    """hscProve :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> m HscProof with the reference's own signature (Signature.hs:32-37): polynomials
    that are NOT the s(X,Y) of a circuit -- dense blocks, a single monomial, positive exponents only (where evaluation points may
    be 0), repeated terms -- every element against the literal restatement; hscVerify (host pairings) accepts, and rejects a
    tampered proof, another polynomial and other points"""
    pyr = random.Random(808)
    d = 40
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    g, s = sonic.SRS.new(d, x, alpha), ref.SRS(d, x, alpha)

    def rnd_poly(xr, yr, density):
        p = {}
        for ex in xr:
            for ey in yr:
                if ex != 0 and ey != 0 and pyr.random() < density:       # X^0 / Y^0 would need the omitted g^alpha (SRS.hs:38)
                    p.setdefault(ex, {})[ey] = pyr.randrange(1, R)
        return p
    polys = [rnd_poly(range(-9, 12), range(-7, 6), 0.3), rnd_poly(range(-3, 4), range(-30, 31), 0.5), {5: {-4: 77}},
             rnd_poly(range(1, 20), range(1, 9), 0.4)]
    for k, sXY in enumerate(polys):
        for m in (0, 1, 3):
            yzs = [(pyr.randrange(1, R), pyr.randrange(1, R)) for _ in range(m)]
            u, v = pyr.randrange(1, R), pyr.randrange(1, R)
            got = sonic.hsc_prove_poly(g, sXY, yzs, u, v)
            want = ref.hsc_prove(s, sXY, yzs, u, v)
            assert (got.hscS, got.hscW, got.hscQv, got.hscC, got.hscU, got.hscV) == \
                (want["hscS"], want["hscW"], want["hscQv"], want["hscC"], want["hscU"], want["hscV"]), (k, m)
            assert sonic.hsc_verify_poly(g, sXY, yzs, got)
            if m:
                import dataclasses
                sjp, wjp, qj = got.hscW[0]
                assert not sonic.hsc_verify_poly(g, sXY, yzs, dataclasses.replace(got, hscW=[((sjp + 1) % R, wjp, qj)] + got.hscW[1:]))
                assert not sonic.hsc_verify_poly(g, sXY, [((yzs[0][0] + 1) % R, yzs[0][1])] + yzs[1:], got)
                other = {ex: dict(inner) for ex, inner in sXY.items()}
                ex0 = next(iter(other))
                ey0 = next(iter(other[ex0]))
                other[ex0][ey0] = (other[ex0][ey0] + 1) % R
                assert not sonic.hsc_verify_poly(g, other, yzs, got)
    # the list form, with a repeated exponent pair (summed) and terms out of order
    terms = [(3, 2, 5), (-2, 1, 9), (3, 2, 6), (1, -1, 4)]
    got = sonic.hsc_prove_poly(g, terms, [(11, 13)], 17, 19)
    want = ref.hsc_prove(s, {3: {2: 11}, -2: {1: 9}, 1: {-1: 4}}, [(11, 13)], 17, 19)
    assert (got.hscS, got.hscW, got.hscQv, got.hscC) == (want["hscS"], want["hscW"], want["hscQv"], want["hscC"])
    # zero as an evaluation point: fine without negative powers of that variable (z_j = 0: the opening is a shift), an error with them
    pos = polys[3]
    got = sonic.hsc_prove_poly(g, pos, [(7, 0)], 3, 5)
    want = ref.hsc_prove(s, pos, [(7, 0)], 3, 5)
    assert (got.hscS, got.hscW, got.hscQv, got.hscC) == (want["hscS"], want["hscW"], want["hscQv"], want["hscC"])
    assert sonic.hsc_verify_poly(g, pos, [(7, 0)], got)
    with pytest.raises(sonic.SonicError) as e:
        sonic.hsc_prove_poly(g, polys[0], [(7, 0)], 3, 5)
    assert e.value.code == 4
    # a term that needs an SRS element beyond d, and the omitted g^alpha (an X^0 term of s(X, y_j)): index errors as in the reference
    with pytest.raises(sonic.SonicError) as e:
        sonic.hsc_prove_poly(g, {d + 1: {1: 1}}, [(2, 3)], 5, 7)
    assert e.value.code == 2
    with pytest.raises(sonic.SonicError) as e:
        sonic.hsc_prove_poly(g, {0: {1: 1}, 2: {1: 1}}, [(2, 3)], 5, 7)
    assert e.value.code == 2
    # the s(X,Y) of a circuit through the generic entry point == the circuit entry point
    circ, asg = ref.rnd_circuit(pyr, 4, 3)
    yzs = [(pyr.randrange(1, R), pyr.randrange(1, R)) for _ in range(2)]
    a = sonic.hsc_prove_poly(g, ref.s_poly(*circ[:3]), yzs, 21, 23)
    b = sonic.hsc_prove(g, sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), circ[3]), yzs, 21, 23)
    assert a == b


def test_product_verifier(sonic, ref, srs_pair):
    """verify . prove inside the product (test/Test/Protocol.hs:14-23): sonic_verify / sonic_pc_v (host pairings over the
    GPU-generated G2 elements) accept GPU proofs, reject tampered ones, and agree with the oracle's verifier"""
    from oracle import pairing as pg
    pyr = random.Random(41)
    n, Q = 3, 2
    circ, asg = ref.rnd_circuit(pyr, n, Q)
    d = 7 * n + 2
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    g = sonic.SRS.new(d, x, alpha)
    circuit = sonic.ArithCircuit(sonic.GateWeights(*circ[:3]), circ[3])
    proof, ro = sonic.prove(g, sonic.Assignment(*asg), circuit, rng=pyr)
    assert sonic.verify(g, circuit, proof, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
    raw = bytearray(proof.to_bytes())
    raw[2 * 96] ^= 1                                   # prA
    bad = sonic.Proof.from_bytes(bytes(raw), Q)
    assert not sonic.verify(g, circuit, bad, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
    assert not sonic.verify(g, circuit, proof, ro.rndOracleY, (ro.rndOracleZ + 1) % R, ro.rndOracleYZs)
    # the Proof object is what gets verified: edited fields are serialised, not a cached copy of the prover's bytes
    import dataclasses
    assert not sonic.verify(g, circuit, dataclasses.replace(proof, prB=(proof.prB + 1) % R), ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
    assert sonic.verify(g, circuit, dataclasses.replace(proof, prB=proof.prB), ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
    # shapes are checked before the C side reads 64 Q bytes of yzs / sonic_proof_size(Q) bytes of proof
    with pytest.raises(ValueError):
        sonic.verify(g, circuit, proof, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs[:1])
    short = dataclasses.replace(proof, prHscProof=dataclasses.replace(proof.prHscProof, hscS=proof.prHscProof.hscS[:1], hscW=proof.prHscProof.hscW[:1]))
    assert not sonic.verify(g, circuit, short, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
    # proof points must lie in the order-r subgroup: the cofactor point (0, 2) is on the curve and is refused
    with pytest.raises(sonic.SonicError) as e:
        sonic.verify(g, circuit, dataclasses.replace(proof, prWa=(0, 2)), ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
    assert e.value.code == 3
    with pytest.raises(sonic.SonicError) as e:
        sonic.verify(g, circuit, dataclasses.replace(proof, prA=R + 5), ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)   # non-canonical Fr
    assert e.value.code == 3
    # pcV on r(X,1) with max = n (test/Test/CommitmentScheme.hs:58-71), against the oracle's pairing check
    rX1 = ref.eval_y(1, ref.r_poly(*asg))
    z = pyr.randrange(1, R)
    F = sonic.commit_poly(g, n, rX1)
    op = sonic.open_poly(g, z, rX1)
    vs = pg.SRS(d, x, alpha)
    assert sonic.pc_v(g, n, F, z, op) and pg.pc_v(vs, n, F, z, op)
    assert not sonic.pc_v(g, n, F, z, ((op[0] + 1) % R, op[1])) and not sonic.pc_v(g, d, F, z, op)


def test_commit_open_degenerate_inputs(sonic, orc, ref, srs_pair):
    """empty / zero / constant / single-term polynomials through commitPoly and openPoly: the normalised sparse form
    of the reference has no zero coefficients, so these reduce to mempty and to an empty quotient"""
    d, x, alpha, g, o = srs_pair
    s = ref.SRS(d, x, alpha)
    assert sonic.commit_poly(g, d, {}) is None                          # fold over no terms = mempty
    assert sonic.commit_poly(g, d, {3: 0, -2: 0}) is None
    assert sonic.commit_poly(g, d, [(5, 7), (5, R - 7)]) is None         # cancels to the zero polynomial
    fz, W = sonic.open_poly(g, 12345, {})
    assert fz == 0 and W is None
    fz, W = sonic.open_poly(g, 12345, {0: 99})                           # constant: f - f(z) = 0
    assert fz == 99 and W is None
    for f in ({1: 5}, {-1: 5}, {d: 3}, {-d: 3}, {-3: 1, 4: R - 1}):
        for maxm in (d, d - 1):
            try:
                want = ref.commit_poly(s, maxm, f)
            except IndexError:
                with pytest.raises(sonic.SonicError) as e:
                    sonic.commit_poly(g, maxm, f)
                assert e.value.code == 2
                continue
            assert sonic.commit_poly(g, maxm, f) == want
        z = 987654321
        try:
            want = ref.open_poly(s, z, f)
        except IndexError:
            with pytest.raises(sonic.SonicError):
                sonic.open_poly(g, z, f)
            continue
        assert sonic.open_poly(g, z, f) == want


def test_open_poly_at_zero(sonic, ref, srs_pair):
    """openPoly at z = 0 of a polynomial without negative exponents: f(0) = c_0 and W = Commit_plain((f - c_0)/X)
    (CommitmentScheme.hs:43-48); with negative exponents `eval` would need 0^-1"""
    d, x, alpha, g, o = srs_pair
    s = ref.SRS(d, x, alpha)
    fz, W = sonic.open_poly(g, 0, {1: 5})
    assert fz == 0 and W == ref.g1_mul(ref.G1_GEN, 5)                    # (5X - 0)/X = 5 -> g^5
    pyr = random.Random(11)
    for f in ({0: 7}, {0: 7, 1: 3}, {3: 9}, {0: 1, 1: 2, 2: 3, 40: R - 1}, _sparse_poly(pyr, 0, 300), _sparse_poly(pyr, 1, 90)):
        assert sonic.open_poly(g, 0, f) == ref.open_poly(s, 0, f)
    with pytest.raises(sonic.SonicError) as e:
        sonic.open_poly(g, 0, {0: 1, d + 2: 5})                           # quotient term X^{d+1}: past gPositiveX
    assert e.value.code == 2
    fz, W = sonic.open_poly(g, 0, {0: 1, d + 1: 5})                       # X^d is the last element: fine
    assert (fz, W) == ref.open_poly(s, 0, {0: 1, d + 1: 5})


def test_msm_lanes_stream(sonic, orc, srs_pair):
    """sonic_msm_submit / sonic_msm_collect: MSMs streamed over two lanes equal the blocking entry point and the oracle; a lane
    holds one MSM at a time; non-canonical scalars surface at collect"""
    from sonic_amd import _lib
    from sonic_amd.commitment import msm_g1_srs
    d, _, _, g, o = srs_pair
    L = _lib.lib()
    rng = np.random.default_rng(9)
    jobs = [(0, -700, 1400), (1, 1, 900), (0, -d, 2 * d + 1), (1, -50, 50), (0, 3, 1), (0, 0, 0)]
    bufs = []
    for basis, e0, n in jobs:
        sc = rand_fr_array(rng, max(n, 1))[:n]
        dp = C.c_void_p()
        _lib.check(L.sonic_dev_alloc(32 * max(n, 1), C.byref(dp)))
        if n:
            _lib.check(L.sonic_dev_upload(dp, sc.ctypes.data, 32 * n))
        bufs.append((sc, dp))
    lanes = [sonic.MsmLane(), sonic.MsmLane()]
    got = []
    lanes[0].submit(g, *jobs[0][:2], bufs[0][1], jobs[0][2])
    for i in range(len(jobs)):
        if i + 1 < len(jobs):
            lanes[(i + 1) & 1].submit(g, *jobs[i + 1][:2], bufs[i + 1][1], jobs[i + 1][2])
        got.append(lanes[i & 1].collect())
    for (basis, e0, n), (sc, dp), out in zip(jobs, bufs, got):
        want = orc.msm_srs(o, basis, e0, sc, 1, NCPU) if n else bytes(96)
        assert out == want == msm_g1_srs(g, basis, e0, sc)
    # partial + sum_partials == the normalised result
    lanes[0].submit(g, 0, -700, bufs[0][1], 1400)
    part = np.frombuffer(lanes[0].collect(partial=True), np.uint8)
    from sonic_amd import distributed as sd
    assert sd.sum_partials(part, 1) == got[0]
    with pytest.raises(sonic.SonicError) as e:
        lanes[0].collect()
    assert e.value.code == 7
    lanes[0].submit(g, 0, -700, bufs[0][1], 1400)
    with pytest.raises(sonic.SonicError) as e:
        lanes[0].submit(g, 0, -700, bufs[0][1], 1400)
    assert e.value.code == 7
    assert lanes[0].collect() == got[0]
    with pytest.raises(sonic.SonicError) as e:
        lanes[0].submit(g, 0, d, bufs[0][1], 2)               # runs past gPositiveX
    assert e.value.code == 2
    bad = np.full((4, 32), 0xFF, np.uint8)
    _lib.check(L.sonic_dev_upload(bufs[0][1], bad.ctypes.data, 128))
    lanes[1].submit(g, 0, -700, bufs[0][1], 4)
    with pytest.raises(sonic.SonicError) as e:
        lanes[1].collect()
    assert e.value.code == 3
    for ln in lanes:
        ln.close()
    for _, dp in bufs:
        L.sonic_dev_free(dp)


def test_msm_zero_and_identity_scalars(sonic, orc, srs_pair):
    from sonic_amd.commitment import msm_g1_srs
    d, _, _, g, o = srs_pair
    n = 3000
    zeros = fr_bytes([0] * n)
    assert msm_g1_srs(g, 0, -1500, zeros) == bytes(96)
    one_hot = fr_bytes([0] * 1234 + [1] + [0] * (n - 1235))
    assert msm_g1_srs(g, 0, -1500, one_hot) == o.points(0, -1500 + 1234, 1)[0].tobytes()
    minus = fr_bytes([0] * 10 + [R - 1] + [0] * (n - 11))
    got = sonic.g1_from_bytes(msm_g1_srs(g, 0, -1500, minus))
    p = sonic.g1_from_bytes(o.points(0, -1490, 1)[0].tobytes())
    Qm = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    assert got == (p[0], (Qm - p[1]) % Qm)


def test_c99_abi_harness(sonic, tmp_path):
    """tests/host/abi_harness.c: examples/Main.hs (prove + verify), the resident handle, the Fiat-Shamir mode and the d < 7n error
    through nothing but include/sonic_hip.h from plain C99 -- what a Haskell `foreign import ccall` shim binds"""
    import subprocess
    from test_abi import _build_harness
    out = subprocess.run([_build_harness(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "abi_harness: OK" in out.stdout, out.stdout + out.stderr


def test_from_x_from_y_lifts(sonic, ref):
    """fromX / fromY (Utils.hs:23-27) lift a univariate polynomial into the bivariate ring; the reference uses them inside tPoly
    (Constraints.hs:56-65).  Here: the lifts of the product's mirror equal the restatement's; hscProve on a lifted polynomial
    fails exactly as the reference does (evalX u / evalY y_j of a lift has a constant term, and commitPoly with max = d needs the
    omitted g^alpha for it: CommitmentScheme.hs:70-73 via SRS.hs:38); hscProve on fromX p + fromY q with the constant terms
    arranged to cancel equals the restatement's element by element and verifies; and tPoly built the reference's way -- fromX (evalY 1 r)
    * (r + s) + fromY (-k) in the bivariate ring, then evalY y -- commits to the T of the GPU's univariate prover."""
    pyr = random.Random(1212)
    d = 64
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    g, s = sonic.SRS.new(d, x, alpha), ref.SRS(d, x, alpha)
    p = {e: pyr.randrange(1, R) for e in (-7, -3, -1, 2, 5, 9)}
    q = {e: pyr.randrange(1, R) for e in (-6, -2, 1, 3, 4)}
    assert sonic.from_x(p) == ref.from_x(p) and sonic.from_y(q) == ref.from_y(q)
    assert sonic.from_x({3: 0, 4: 5}) == {4: {0: 5}} and sonic.from_y({}) == {} and sonic.from_y({2: 0}) == {}
    yz, u, v = [(pyr.randrange(1, R), pyr.randrange(1, R))], pyr.randrange(1, R), pyr.randrange(1, R)
    # a bare lift: the reference's `index` panics on the hole of the alpha basis; both sides refuse
    for lifted in (sonic.from_x(p), sonic.from_y(q)):
        with pytest.raises(sonic.SonicError) as e:
            sonic.hsc_prove_poly(g, lifted, yz, u, v)
        assert e.value.code == 2
        with pytest.raises(Exception):
            ref.hsc_prove(s, lifted, yz, u, v)
    # fromX p + fromY q with p_0, q_0 such that s(X, y_1) and s(u, Y) have no constant term:
    #   p_0 + q(y_1) = 0 and q_0 + p(u) = 0  <=>  p'(u) = q'(y_1) (primes: without the constant terms) and p_0 + q_0 = -p'(u)
    ev = lambda f, t: sum(c * pow(t, e, R) for e, c in f.items()) % R        # noqa: E731
    (y1, _z1), = yz
    scale = ev(p, u) * pow(ev(q, y1), -1, R) % R
    q = {e: c * scale % R for e, c in q.items()}
    assert ev(p, u) == ev(q, y1)
    p0 = pyr.randrange(1, R)
    p[0] = p0
    q[0] = (-ev({e: c for e, c in p.items() if e}, u) - p0) % R
    sXY = sonic.biv_add(sonic.from_x(p), sonic.from_y(q))
    assert sXY == ref.biv_add(ref.from_x(p), ref.from_y(q))
    got = sonic.hsc_prove_poly(g, sXY, yz, u, v)
    want = ref.hsc_prove(s, sXY, yz, u, v)
    assert (got.hscS, got.hscW, got.hscQv, got.hscC, got.hscU, got.hscV) == \
        (want["hscS"], want["hscW"], want["hscQv"], want["hscC"], want["hscU"], want["hscV"])
    assert sonic.hsc_verify_poly(g, sXY, yz, got)
    # tPoly the reference's way (lifts and a bivariate product), against the T of the GPU prover
    n, Q = 3, 2
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
    proof, _ = sonic.prove(g, sonic.Assignment(*asg), sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3]), transcript=tr)
    rXY = ref.r_poly(*asg)
    for i, c in enumerate(tr[:4], start=1):                                    # the blinders, Protocol.hs:58-62
        rXY = ref.biv_add(rXY, {-2 * n - i: {-2 * n - i: c}})
    sXY_c = ref.s_poly(circ[0], circ[1], circ[2])
    kY = ref.k_poly(circ[3], n)
    rX1 = ref.from_x(ref.eval_y(1, rXY))
    tXY = ref.biv_add(ref.biv_mul(rX1, ref.biv_add(rXY, sXY_c)), ref.from_y(ref.lp_neg(kY)))
    assert tXY == ref.t_poly(rXY, sXY_c, kY)
    assert sonic.commit_poly(g, d, ref.eval_y(tr[4], tXY)) == proof.prT
