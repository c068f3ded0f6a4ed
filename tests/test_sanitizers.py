"""AddressSanitizer + UndefinedBehaviorSanitizer over the product's HOST code (CPU only -- GPU sanitizers are not
available on this pool): the limb arithmetic / point formulas as the host compiles them, the shared-inversion normalisations, the
SHA-256 and Fiat-Shamir transcript code, the SRS file parser on hostile files (tests/host/san_host.cpp); the verifier's whole host
path -- sonic_pc_v, sonic_hsc_verify, sonic_verify, sonic_verify_fs from sonic_amd/csrc/verify.hip compiled as C++ -- on a golden
proof and tampered copies of it (tests/host/san_verify.cpp); and the pairing self-test (tests/pairing_selftest.cpp).  Any report
aborts the binary (-fno-sanitize-recover=all)."""
import json
import os
import struct
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(HERE, "host")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", HOST, "-s", "-j3", "san"])
    return HOST


def srs_pairing_bytes(alpha: int) -> bytes:
    """srsPairing = e(g, h^alpha) (SRS.hs:21,42) from the python oracle, in the layout of sonic_srs_pairing (include/sonic_hip.h): the
    reduced ate pairing for the NEGATIVE curve parameter is the inverse of the oracle's f_{|x|}^((q^12-1)/r); its polynomial-basis
    coefficients over Fq[w]/(w^12 - 2 w^6 + 2) become the tower's Fq2 coefficients through u = w^6 - 1: the coefficient of w^i is
    (c_i + c_{i+6}) + c_{i+6} u, and w^i sits at [i % 2][i // 2] of Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - (1 + u))"""
    from oracle import pairing as pr
    from oracle.sonic_ref import G1_GEN, Q as FQ
    e = pr.f12_inv(pr.pairing(G1_GEN, pr.g2_mul(pr.G2_GEN, alpha)))
    out = b""
    for i in range(2):
        for j in range(3):
            k = 2 * j + i
            out += ((e[k] + e[k + 6]) % FQ).to_bytes(48, "little") + (e[k + 6] % FQ).to_bytes(48, "little")
    return out


def _fr(hexes):
    return b"".join(int(h, 16).to_bytes(32, "little") for h in hexes)


def test_host_arithmetic_transcript_and_srs_file_parser(built, tmp_path):
    out = subprocess.run([os.path.join(built, "san_host"), str(tmp_path)], capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode == 0 and "san_host ok" in out.stdout, out.stdout + out.stderr[-3000:]


@pytest.mark.parametrize("name", ["example1", "rnd_n3"])
def test_verifier_host_path(built, tmp_path, name):
    c = next(x for x in json.load(open(os.path.join(HERE, "golden", "prove_small.json")))["cases"] if x["name"] == name)
    n, Q, d = c["n"], c["Q"], c["d"]
    tr = c["transcript"]
    flat = lambda w: _fr([v for r in w for v in r])    # noqa: E731
    yzs = b"".join(_fr([tr[6 + j]]) + _fr([tr[6 + Q + j]]) for j in range(Q))
    blob = struct.pack("<qqq", n, Q, d) + _fr([c["x"], c["alpha"]]) + flat(c["wL"]) + flat(c["wR"]) + flat(c["wO"]) + _fr(c["cs"]) + \
        bytes.fromhex(c["proof"]) + _fr([tr[4], tr[5]]) + yzs + srs_pairing_bytes(int(c["alpha"], 16))
    path = tmp_path / "case.bin"
    path.write_bytes(blob)
    out = subprocess.run([os.path.join(built, "san_verify"), str(path)], capture_output=True, text=True, timeout=900, env=ENV)
    assert out.returncode == 0 and "san_verify ok" in out.stdout, out.stdout + out.stderr[-3000:]


def test_pairing_selftest_under_sanitizers(built):
    out = subprocess.run([os.path.join(built, "san_pairing")], capture_output=True, text=True, timeout=900, env=ENV)
    assert out.returncode == 0 and "pairing selftest ok" in out.stdout, out.stdout + out.stderr[-3000:]
