"""The opt-in Fiat-Shamir transcript without a GPU: the python restatement (oracle/sonic_ref.py, hashlib) against the committed
fixture, and the product's own SHA-256 / transcript code (sonic_amd/csrc/fs.hpp, host functions of libsonic_hip.so that touch no
device) against the python on the same bytes."""
import ctypes as C
import hashlib
import json
import os

import numpy as np

from util import R, fr_bytes

HERE = os.path.dirname(os.path.abspath(__file__))
FS = json.load(open(os.path.join(HERE, "golden", "fs_small.json")))["cases"]
BASE = {c["name"]: c for c in json.load(open(os.path.join(HERE, "golden", "prove_small.json")))["cases"]}


def _case(c):
    b = BASE[c["name"]]
    iv = lambda v: int(v, 16)    # noqa: E731
    circ = ([[iv(v) for v in r] for r in b["wL"]], [[iv(v) for v in r] for r in b["wR"]], [[iv(v) for v in r] for r in b["wO"]], [iv(v) for v in b["cs"]])
    asg = ([iv(v) for v in b["aL"]], [iv(v) for v in b["aR"]], [iv(v) for v in b["aO"]])
    return b, circ, asg


def test_python_fs_matches_fixture(ref):
    c = FS[0]                                    # example1 (the others run on the GPU suite; python proving is slow)
    b, circ, asg = _case(c)
    assert ref.fs_circuit_digest(circ).hex() == c["circuit_digest"]
    srs = ref.SRS(b["d"], int(b["x"], 16), int(b["alpha"], 16))
    assert ref.fs_srs_id(srs).hex() == c["srs_id"] and ref.fs_witness_digest(asg).hex() == c["witness_digest"]
    proof, _o, tr = ref.prove_fs(srs, asg, circ, bytes.fromhex(c["seed"]))
    assert ref.proof_to_bytes(proof).hex() == c["proof"] and ["%x" % v for v in tr] == c["transcript"]


def test_fixture_is_a_fixed_point(ref):
    """every challenge of the fixture's transcript is the hash of the fixture's proof bytes up to its draw site"""
    for c in FS:
        b, circ, asg = _case(c)
        pb = bytes.fromhex(c["proof"])
        dg, sid, wd = (bytes.fromhex(c[k]) for k in ("circuit_digest", "srs_id", "witness_digest"))
        y, z, ys, zs, u, v = ref.fs_challenges_of_proof(b["n"], b["Q"], b["d"], dg, sid, pb)
        tr = [int(t, 16) for t in c["transcript"]]
        assert tr[4:] == [y, z] + ys + zs + [u, v]
        assert tr[:4] == ref.fs_blinders(bytes.fromhex(c["seed"]), dg, sid, wd)
        assert wd == ref.fs_witness_digest(asg)
        # the blinders depend on the statement, the reference string and the witness, not on the seed alone: one seed on
        # another assignment / circuit / SRS gives unrelated blinders
        other = ref.fs_witness_digest((asg[0], asg[1], [(v_ + 1) % R for v_ in asg[2]]))
        for alt in (ref.fs_blinders(bytes.fromhex(c["seed"]), dg, sid, other), ref.fs_blinders(bytes.fromhex(c["seed"]), bytes(32), sid, wd),
                    ref.fs_blinders(bytes.fromhex(c["seed"]), dg, bytes(32), wd)):
            assert not set(alt) & set(tr[:4])
        # ... and the challenges on the reference string, not only on its degree
        assert ref.fs_challenges_of_proof(b["n"], b["Q"], b["d"], dg, bytes(32), pb)[0] != y
        assert pb[-64:-32] == ref.fr_to_bytes(u) and pb[-32:] == ref.fr_to_bytes(v)


def test_product_transcript_code_matches_python(ref):
    from sonic_amd import _lib
    L = _lib.lib()
    for c in FS:
        b, circ, asg = _case(c)
        wL, wR, wO, cs = circ
        flat = lambda w: fr_bytes([v for r in w for v in r])    # noqa: E731
        out = C.create_string_buffer(32)
        aL, aR, aO, acs = flat(wL), flat(wR), flat(wO), fr_bytes(cs)
        assert L.sonic_fs_circuit_digest(b["n"], b["Q"], aL.ctypes.data, aR.ctypes.data, aO.ctypes.data, acs.ctypes.data, out) == 0
        assert out.raw.hex() == c["circuit_digest"]
        ch = C.create_string_buffer(32 * (4 + 2 * b["Q"]))
        assert L.sonic_fs_challenges_v2(b["n"], b["Q"], b["d"], out.raw, bytes.fromhex(c["srs_id"]), bytes.fromhex(c["proof"]), ch) == 0
        got = [int.from_bytes(ch.raw[32 * i:32 * i + 32], "little") for i in range(4 + 2 * b["Q"])]
        assert got == [int(t, 16) for t in c["transcript"]][4:]


def test_sha256_known_answers():
    """FIPS 180-4 vectors through the product's SHA-256, reached via the circuit digest: SHA256(label || le64 n || le64 Q || data)
    for data lengths that cross the 55 / 56 / 64-byte padding boundaries"""
    from sonic_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 5, 17):
        Q = 1
        w = [rng.integers(0, 256, size=(n, 32), dtype=np.uint8) for _ in range(3)]
        for a in w:
            a[:, 31] &= 0x3f
        cs = rng.integers(0, 256, size=(Q, 32), dtype=np.uint8)
        cs[:, 31] &= 0x3f
        out = C.create_string_buffer(32)
        assert L.sonic_fs_circuit_digest(n, Q, w[0].ctypes.data, w[1].ctypes.data, w[2].ctypes.data, cs.ctypes.data, out) == 0
        want = hashlib.sha256(b"sonic-hip/circuit/v1" + n.to_bytes(8, "little") + Q.to_bytes(8, "little") +
                              w[0].tobytes() + w[1].tobytes() + w[2].tobytes() + cs.tobytes()).digest()
        assert out.raw == want
