"""CPU unit tests of the product's limb arithmetic and point formulas (sonic_amd/csrc/field.hpp,
g1.hpp compiled for the host by tests/host/Makefile) against python big integers."""
import ctypes as C
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def L():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), "-s"])
    return C.CDLL(os.path.join(HERE, "host", "libfield_host.so"))


def _fop(fn, nb, op, a, b=0):
    out = C.create_string_buffer(nb)
    assert fn(op, a.to_bytes(nb, "little"), b.to_bytes(nb, "little"), out) == 0
    return int.from_bytes(out.raw, "little")


@pytest.mark.parametrize("field", ["fq", "fr"])
def test_field_ops(L, ref, field):
    fn, mod, nb = (L.host_fq_op, ref.Q, 48) if field == "fq" else (L.host_fr_op, ref.R, 32)
    rng = random.Random(5)
    vals = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (1 << (nb * 8 - 3)) % mod] + [rng.randrange(mod) for _ in range(150)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        assert _fop(fn, nb, 0, a, b) == a * b % mod
        assert _fop(fn, nb, 1, a, b) == (a + b) % mod
        assert _fop(fn, nb, 2, a, b) == (a - b) % mod
        assert _fop(fn, nb, 3, a) == (-a) % mod
        if a and i < 25:
            assert _fop(fn, nb, 4, a) == pow(a, -1, mod)


def test_g1_formulas(L, ref):
    """XYZZ mixed add / full add / doubling incl. P+P, P+(-P), infinity operands"""
    rng = random.Random(6)

    def gop(op, p, q, k=0):
        out = C.create_string_buffer(96)
        assert L.host_g1_op(op, ref.g1_to_bytes(p), ref.g1_to_bytes(q), k, out) == 0
        return ref.g1_from_bytes(out.raw)

    out = C.create_string_buffer(96)
    L.host_gen(out)
    assert ref.g1_from_bytes(out.raw) == ref.G1_GEN
    G = ref.G1_GEN
    pts = [ref.g1_mul(G, rng.randrange(1, ref.R)) for _ in range(4)] + [ref.INF, G]
    for p in pts:
        for q in pts + [ref.g1_neg(p), p]:
            assert gop(0, p, q) == ref.g1_add(p, q)
            assert gop(1, p, q) == ref.g1_add(ref.g1_mul(p, 4) if p else None, ref.g1_mul(q, 3) if q else None)
            assert gop(2, p, q) == ref.g1_add(p, p)
            assert gop(4, p, q) == p
        for k in [0, 1, 2, 3, 17, 65535, 32768]:
            assert gop(3, p, p, k) == (ref.g1_mul(p, 3 * k) if p else None)


def test_endomorphism_split_and_phi(L, ref):
    """endo.hpp on the host: s = s1 + lambda s2 with both halves below 2^128 for edge and random scalars, and phi(P) = (beta x, y)
    equals lambda P (python big integers / the literal restatement's group law)"""
    z = 0xd201000000010000
    lam = z * z - 1
    assert (lam * lam + lam + 1) % ref.R == 0
    rng = random.Random(8)
    vals = [0, 1, lam - 1, lam, lam + 1, 2 * lam - 1, 2 * lam, ref.R - 1, ref.R - lam, (ref.R - 1) // 2, lam * lam, lam * (lam + 1) - 1,
            (1 << 128) - 1, 1 << 128, (1 << 254) + 12345] + [rng.randrange(ref.R) for _ in range(3000)]
    for s in vals:
        a, b = C.create_string_buffer(32), C.create_string_buffer(32)
        assert L.host_endo_split(s.to_bytes(32, "little"), a, b) == 0
        s1, s2 = int.from_bytes(a.raw, "little"), int.from_bytes(b.raw, "little")
        assert (s1, s2) == (s % lam, s // lam) and s1 < 1 << 128 and s2 < 1 << 128, hex(s)
    for k in (1, 2, 12345, rng.randrange(1, ref.R)):
        P = ref.g1_mul(ref.G1_GEN, k)
        out = C.create_string_buffer(96)
        assert L.host_endo_phi(ref.g1_to_bytes(P), out) == 0
        assert ref.g1_from_bytes(out.raw) == ref.g1_mul(P, 2 * lam % ref.R)        # phi(2P) = lambda 2P
