"""Host logic of the verifier, no GPU: the tower-field pairing of sonic_amd/csrc/pairing.hpp (what pcV / verify / hscVerify
evaluate, CommitmentScheme.hs:58-68) and the host's 64-bit-limb field product, compiled with g++ from the product's own headers
and checked by tests/pairing_selftest.cpp against the plain polynomial-basis pairing (tests/pairing_plain.hpp), bilinearity,
and a pcV-shaped product with a known trapdoor."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
X = -0xd201000000010000


def test_final_exponent_identity():
    """the exponent pairing.hpp applies: 3 (q^12 - 1)/r = (q^6 - 1)(q^2 + 1) [(x-1)^2 (x+q)(x^2+q^2-1) + 3]; 3 is prime to r, and
    q^4 - 1 divides it (line factors in Fq2 and Fq4 vanish)"""
    assert (Q ** 4 - Q ** 2 + 1) % R == 0
    hard3 = (X - 1) ** 2 * (X + Q) * (X * X + Q * Q - 1) + 3
    assert hard3 == 3 * ((Q ** 4 - Q ** 2 + 1) // R)
    full = (Q ** 6 - 1) * (Q ** 2 + 1) * hard3
    assert full == 3 * ((Q ** 12 - 1) // R) and full % (Q ** 4 - 1) == 0 and R % 3 != 0
    assert (Q - 1) % 6 == 0                      # the Frobenius constants xi^((q-1)/6)


def test_pairing_selftest(tmp_path):
    exe = str(tmp_path / "pairing_selftest")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "sonic_amd", "csrc"), "-I", os.path.join(ROOT, "tests"),
                    os.path.join(ROOT, "tests", "pairing_selftest.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "pairing selftest ok" in out.stdout, out.stdout + out.stderr
