"""ctypes binding of oracle/libsonic_oracle.so (the plain-C CPU oracle).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by anything under sonic_amd/.

All values cross the boundary in the canonical encodings of include/sonic_hip.h:
Fr = 32 B little-endian < r, G1 = 96 B x||y little-endian (all-zero = infinity).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libsonic_oracle.so")
    src = os.path.join(_HERE, "sonic_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libsonic_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_srs_new.restype = C.c_void_p
        L.orc_srs_new.argtypes = [C.c_long, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_srs_from_points.restype = C.c_void_p
        L.orc_srs_from_points.argtypes = [C.c_long, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_srs_free.argtypes = [C.c_void_p]
        L.orc_srs_d.restype = C.c_long
        L.orc_srs_d.argtypes = [C.c_void_p]
        L.orc_srs_points.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_void_p]
        L.orc_msm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int]
        L.orc_msm_srs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int]
        L.orc_commit_poly.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_open_poly.argtypes = [C.c_void_p, C.c_char_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ntt.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_poly_mul.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_int]
        L.orc_proof_size.restype = C.c_long
        L.orc_proof_size.argtypes = [C.c_long]
        L.orc_prove.argtypes = [C.c_void_p, C.c_long, C.c_long] + [C.c_void_p] * 8 + [C.c_void_p, C.c_int]
        L.orc_set_mode.argtypes = [C.c_int, C.c_int]
        L.orc_g1_mul.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_g1_add.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_g1_gen.argtypes = [C.c_void_p]
        L.orc_g1_on_curve.argtypes = [C.c_char_p]
        L.orc_fr_mul.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_fr_inv.argtypes = [C.c_void_p, C.c_char_p]
    return _LIB


class OracleError(Exception):
    def __init__(self, code):
        super().__init__({1: "D_TOO_SMALL", 2: "SRS_INDEX_OUT_OF_RANGE", 3: "BAD_ENCODING", 4: "INEXACT_DIVISION"}.get(code, str(code)))
        self.code = code


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def fr_bytes(x: int) -> bytes:
    return (x % R).to_bytes(32, "little")


def fr_array(vals) -> np.ndarray:
    """list of ints -> uint8 array [len, 32]"""
    return np.frombuffer(b"".join(fr_bytes(v) for v in vals), dtype=np.uint8).reshape(-1, 32).copy() if len(vals) else np.zeros((0, 32), np.uint8)


def fr_list(arr) -> list:
    b = np.ascontiguousarray(arr, dtype=np.uint8).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if isinstance(a, np.ndarray) else a


class SRS:
    """oracle-side SRS.new (SRS.hs:27-43): basis 0 = g^{x^e}, basis 1 = g^{alpha x^e}, e in [-d, d]."""

    def __init__(self, d: int, x: int, alpha: int, threads: int = 8):
        self.d = d
        self.h = lib().orc_srs_new(d, fr_bytes(x), fr_bytes(alpha), threads)
        if not self.h:
            raise OracleError(3)

    @classmethod
    def from_points(cls, d: int, basis0: np.ndarray, basis1: np.ndarray, threads: int = 8) -> "SRS":
        """the record constructor: canonical bytes, uint8 [(2d+1), 96] per basis (validated: canonical, on the curve)"""
        b0, b1 = np.ascontiguousarray(basis0, np.uint8), np.ascontiguousarray(basis1, np.uint8)
        assert b0.size == 96 * (2 * d + 1) and b1.size == 96 * (2 * d + 1)
        self = cls.__new__(cls)
        self.d = d
        self.h = lib().orc_srs_from_points(d, _p(b0), _p(b1), threads)
        if not self.h:
            raise OracleError(3)
        return self

    def points(self, basis: int, e0: int, n: int) -> np.ndarray:
        out = np.zeros((n, 96), np.uint8)
        _chk(lib().orc_srs_points(self.h, basis, e0, n, _p(out)))
        return out

    def __del__(self):
        try:
            if self.h:
                lib().orc_srs_free(self.h)
                self.h = None
        except Exception:
            pass


def set_mode(msm_mode: int, threads: int):
    lib().orc_set_mode(msm_mode, threads)


def msm(points: np.ndarray, scalars: np.ndarray, mode: int = 1, threads: int = 8) -> bytes:
    points = np.ascontiguousarray(points, np.uint8)
    scalars = np.ascontiguousarray(scalars, np.uint8)
    n = scalars.size // 32
    out = C.create_string_buffer(96)
    _chk(lib().orc_msm(out, _p(points), _p(scalars), n, mode, threads))
    return out.raw


def msm_srs(srs: SRS, basis: int, e0: int, scalars: np.ndarray, mode: int = 1, threads: int = 8) -> bytes:
    scalars = np.ascontiguousarray(scalars, np.uint8)
    n = scalars.size // 32
    out = C.create_string_buffer(96)
    _chk(lib().orc_msm_srs(out, srs.h, basis, e0, _p(scalars), n, mode, threads))
    return out.raw


def commit_poly(srs: SRS, maxm: int, exps, coeffs: np.ndarray) -> bytes:
    exps = np.ascontiguousarray(exps, np.int64)
    coeffs = np.ascontiguousarray(coeffs, np.uint8)
    out = C.create_string_buffer(96)
    _chk(lib().orc_commit_poly(srs.h, maxm, len(exps), _p(exps), _p(coeffs), out))
    return out.raw


def open_poly(srs: SRS, z: int, exps, coeffs: np.ndarray):
    exps = np.ascontiguousarray(exps, np.int64)
    coeffs = np.ascontiguousarray(coeffs, np.uint8)
    out = C.create_string_buffer(96)
    fz = C.create_string_buffer(32)
    _chk(lib().orc_open_poly(srs.h, fr_bytes(z), len(exps), _p(exps), _p(coeffs), fz, out))
    return int.from_bytes(fz.raw, "little"), out.raw


def ntt(data: np.ndarray, inverse: bool = False) -> np.ndarray:
    a = np.ascontiguousarray(data, np.uint8).copy()
    n = a.size // 32
    logn = n.bit_length() - 1
    assert 1 << logn == n
    _chk(lib().orc_ntt(_p(a), logn, int(inverse)))
    return a


def poly_mul(a: np.ndarray, b: np.ndarray, use_ntt: bool = True) -> np.ndarray:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    na, nb = a.size // 32, b.size // 32
    out = np.zeros((na + nb - 1, 32), np.uint8)
    _chk(lib().orc_poly_mul(_p(a), na, _p(b), nb, _p(out), int(use_ntt)))
    return out


def proof_size(Q: int) -> int:
    return lib().orc_proof_size(Q)


def prove(srs: SRS, n: int, Q: int, wL, wR, wO, cs, aL, aR, aO, transcript, use_ntt: bool = True) -> bytes:
    """all vector arguments: uint8 arrays of canonical Fr (weights dense Q x n row-major)."""
    arrs = [np.ascontiguousarray(a, np.uint8) for a in (wL, wR, wO, cs, aL, aR, aO, transcript)]
    out = C.create_string_buffer(proof_size(Q))
    _chk(lib().orc_prove(srs.h, n, Q, *[_p(a) for a in arrs], out, int(use_ntt)))
    return out.raw


def g1_mul(p96: bytes, k: int) -> bytes:
    out = C.create_string_buffer(96)
    _chk(lib().orc_g1_mul(out, p96, fr_bytes(k)))
    return out.raw


def g1_add(p96: bytes, q96: bytes) -> bytes:
    out = C.create_string_buffer(96)
    _chk(lib().orc_g1_add(out, p96, q96))
    return out.raw


def g1_gen() -> bytes:
    out = C.create_string_buffer(96)
    lib().orc_g1_gen(out)
    return out.raw


def g1_on_curve(p96: bytes) -> bool:
    return bool(lib().orc_g1_on_curve(p96))
