"""Sonic.SRS (src/Sonic/SRS.hs): the structured reference string, resident in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .encoding import fr_to_bytes, g1_from_bytes


class SRS:
    """`data SRS` (SRS.hs:11-22), prover half.  Device layout: two arrays of 2d+1 affine points,
    basis 0 = g^{x^e}, basis 1 = g^{alpha x^e}, slot e + d; the reference's four G1 vectors are views:
    gNegativeX[k] = basis0[-(k+1)], gPositiveX[k] = basis0[k], gNegativeAlphaX[k] = basis1[-(k+1)],
    gPositiveAlphaX[k] = basis1[k+1].  The G2 vectors (verifier only) are generated lazily on first access, after which
    the handle forgets x and alpha."""

    def __init__(self, handle: C.c_void_p, d: int):
        self._h = handle
        self.srsD = d

    @classmethod
    def new(cls, d: int, x: int, alpha: int, device: int = -1) -> "SRS":
        """SRS.new :: Int -> Fr -> Fr -> SRS (SRS.hs:27-43), generated on the GPU `device` (-1: the default device)."""
        h = C.c_void_p()
        _lib.check(_lib.lib().sonic_srs_new_on(device, d, fr_to_bytes(x), fr_to_bytes(alpha), C.byref(h)))
        return cls(h, d)

    @classmethod
    def from_points(cls, d: int, basis0: np.ndarray, basis1: np.ndarray, device: int = -1) -> "SRS":
        """The record constructor: caller-supplied points, uint8 [(2d+1), 96] per basis."""
        b0 = np.ascontiguousarray(basis0, np.uint8)
        b1 = np.ascontiguousarray(basis1, np.uint8)
        assert b0.size == 96 * (2 * d + 1) and b1.size == 96 * (2 * d + 1)
        h = C.c_void_p()
        _lib.check(_lib.lib().sonic_srs_from_points_on(device, d, b0.ctypes.data, b1.ctypes.data, C.byref(h)))
        return cls(h, d)

    def replicate(self, device: int) -> "SRS":
        """a replica of this SRS on another GPU of the process, copied device to device with its window tables (sonic_srs_replicate)"""
        h = C.c_void_p()
        _lib.check(_lib.lib().sonic_srs_replicate(self._h, device, C.byref(h)))
        return SRS(h, self.srsD)

    @property
    def device(self) -> int:
        """the GPU the handle lives on"""
        return int(_lib.lib().sonic_srs_device(self._h))

    @property
    def srsPairing(self):
        """srsPairing = e(g, h^alpha) (SRS.hs:21,42): the Fq12 element as nested tuples ((c00, c01, c02), (c10, c11, c12)) of Fq2
        pairs -- Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - (1 + u)) -- from sonic_srs_pairing (host pairing; needs the G2 half)"""
        out = C.create_string_buffer(576)
        _lib.check(_lib.lib().sonic_srs_pairing(self._h, out))
        v = [int.from_bytes(out.raw[48 * i:48 * i + 48], "little") for i in range(12)]
        return tuple(tuple((v[6 * i + 2 * j], v[6 * i + 2 * j + 1]) for j in range(3)) for i in range(2))

    def set_g2_points(self, basis0: np.ndarray, basis1: np.ndarray) -> None:
        """attach the verifier half to a handle built from G1 points: uint8 [(2d+1), 192] per basis, validated"""
        b0 = np.ascontiguousarray(basis0, np.uint8)
        b1 = np.ascontiguousarray(basis1, np.uint8)
        assert b0.size == 192 * (2 * self.srsD + 1) and b1.size == 192 * (2 * self.srsD + 1)
        _lib.check(_lib.lib().sonic_srs_set_g2_points(self._h, b0.ctypes.data, b1.ctypes.data))

    def has_g2(self) -> bool:
        """whether the handle holds the verifier half or can still generate it (sonic_srs_has_g2)"""
        return bool(_lib.lib().sonic_srs_has_g2(self._h))

    def save(self, path: str, g2=None) -> None:
        """write the SRS to disk (format in include/sonic_hip.h): the G1 bases and, with g2, the G2 bases, so that the
        loaded handle can verify as well as prove.  g2=None (default): include the G2 half if the handle has it -- handles made
        from G1 points only, or loaded from a version-1 file, save as they are.  The file never holds the trapdoor."""
        _lib.check(_lib.lib().sonic_srs_save(self._h, str(path).encode(), 2 if g2 is None else (1 if g2 else 0)))

    @classmethod
    def load(cls, path: str, device: int = -1) -> "SRS":
        h = C.c_void_p()
        _lib.check(_lib.lib().sonic_srs_load_on(device, str(path).encode(), C.byref(h)))
        return cls(h, int(_lib.lib().sonic_srs_d(h)))

    def points(self, basis: int, e0: int, n: int) -> np.ndarray:
        out = np.zeros((n, 96), np.uint8)
        _lib.check(_lib.lib().sonic_srs_get_points(self._h, basis, e0, n, out.ctypes.data))
        return out

    def g2_points(self, basis: int, e0: int, n: int) -> np.ndarray:
        """G2 half (generated on the GPU on first use): uint8 [n, 192], x.c0 || x.c1 || y.c0 || y.c1"""
        out = np.zeros((n, 192), np.uint8)
        _lib.check(_lib.lib().sonic_srs_get_g2_points(self._h, basis, e0, n, out.ctypes.data))
        return out

    def _one_g2(self, name, basis, e, k, length):
        if not (0 <= k < length):
            raise IndexError(f"{name} is not long enough: {k} >= {length}")
        b = self.g2_points(basis, e, 1)[0].tobytes()
        if b == bytes(192):
            return None
        v = [int.from_bytes(b[i:i + 48], "little") for i in range(0, 192, 48)]
        return ((v[0], v[1]), (v[2], v[3]))

    def hNegativeX(self, k):            # SRS.hs:35
        return self._one_g2("hNegativeX", 0, -(k + 1), k, self.srsD)

    def hPositiveX(self, k):            # SRS.hs:36
        return self._one_g2("hPositiveX", 0, k, k, self.srsD + 1)

    def hNegativeAlphaX(self, k):       # SRS.hs:40
        return self._one_g2("hNegativeAlphaX", 1, -(k + 1), k, self.srsD)

    def hPositiveAlphaX(self, k):       # SRS.hs:41
        return self._one_g2("hPositiveAlphaX", 1, k, k, self.srsD + 1)

    def _one(self, name, basis, e, k, length):
        if not (0 <= k < length):   # CommitmentScheme.hs:70-73
            raise IndexError(f"{name} is not long enough: {k} >= {length}")
        return g1_from_bytes(self.points(basis, e, 1)[0].tobytes())

    def gNegativeX(self, k):
        return self._one("gNegativeX", 0, -(k + 1), k, self.srsD)

    def gPositiveX(self, k):
        return self._one("gPositiveX", 0, k, k, self.srsD + 1)

    def gNegativeAlphaX(self, k):
        return self._one("gNegativeAlphaX", 1, -(k + 1), k, self.srsD)

    def gPositiveAlphaX(self, k):
        return self._one("gPositiveAlphaX", 1, k + 1, k, self.srsD)

    def close(self):
        if self._h:
            _lib.lib().sonic_srs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
