"""Sonic.CommitmentScheme (src/Sonic/CommitmentScheme.hs): commitPoly / openPoly over the GPU MSM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .encoding import fr_array, fr_to_bytes, g1_from_bytes
from .srs import SRS


def _sparse(poly):
    """VLaurent Fr as {exponent: coeff}, [(exponent, coeff)], or (int64 exps, uint8 coeffs[k,32])."""
    if isinstance(poly, tuple) and len(poly) == 2 and isinstance(poly[0], np.ndarray):
        return np.ascontiguousarray(poly[0], np.int64), fr_array(poly[1])
    items = sorted(poly.items()) if isinstance(poly, dict) else list(poly)
    exps = np.array([e for e, _ in items], dtype=np.int64)
    return exps, fr_array([c for _, c in items])


def commit_poly(srs: SRS, maxm: int, poly):
    """commitPoly :: SRS -> Int -> VLaurent Fr -> G1 (CommitmentScheme.hs:20-33)."""
    exps, coeffs = _sparse(poly)
    out = C.create_string_buffer(96)
    _lib.check(_lib.lib().sonic_commit_poly(srs._h, maxm, len(exps), exps.ctypes.data, coeffs.ctypes.data, out))
    return g1_from_bytes(out.raw)


def open_poly(srs: SRS, z: int, poly):
    """openPoly :: SRS -> Fr -> VLaurent Fr -> (Fr, G1) (CommitmentScheme.hs:36-48)."""
    exps, coeffs = _sparse(poly)
    out = C.create_string_buffer(96)
    fz = C.create_string_buffer(32)
    _lib.check(_lib.lib().sonic_open_poly(srs._h, fr_to_bytes(z), len(exps), exps.ctypes.data, coeffs.ctypes.data, fz, out))
    return int.from_bytes(fz.raw, "little"), g1_from_bytes(out.raw)


def pc_v(srs: SRS, maxm: int, commitment, z: int, opening) -> bool:
    """pcV :: SRS -> Int -> G1 -> Fr -> (Fr, G1) -> Bool (CommitmentScheme.hs:51-68); host-side pairing check"""
    from .encoding import g1_to_bytes
    v, w = opening
    ok = C.c_int(0)
    _lib.check(_lib.lib().sonic_pc_v(srs._h, maxm, g1_to_bytes(commitment), fr_to_bytes(z), fr_to_bytes(v), g1_to_bytes(w), C.byref(ok)))
    return bool(ok.value)


def msm_g1(points: np.ndarray, scalars) -> bytes:
    """The fold inside commitPoly/openPoly on caller-supplied points: sum scalars[i] * points[i].
    points: uint8 [n, 96]; scalars: ints or uint8 [n, 32].  Returns the 96 canonical bytes."""
    pts = np.ascontiguousarray(points, np.uint8)
    sc = fr_array(scalars)
    out = C.create_string_buffer(96)
    _lib.check(_lib.lib().sonic_msm_g1(pts.ctypes.data, sc.ctypes.data, sc.shape[0], out))
    return out.raw


def msm_g1_srs(srs: SRS, basis: int, e0: int, scalars) -> bytes:
    sc = fr_array(scalars)
    out = C.create_string_buffer(96)
    _lib.check(_lib.lib().sonic_msm_g1_srs(srs._h, basis, e0, sc.ctypes.data, sc.shape[0], out))
    return out.raw


def msm_g1_srs_multi(replicas, basis: int, e0: int, scalars, mode: int = 0, d_slices=None) -> bytes:
    """ONE MSM sum_i s_i B[e0 + i] over several SRS replicas -- one per GPU, all in this process (sonic_msm_g1_srs_multi[_dev]).
    mode 0: by term range (every GPU a whole MSM over its slice, the host adds the partial sums); mode 1: by bucket range (every GPU
    accumulates its slice, the GPUs pull their bucket range from each other, each reduces 1/world of the buckets: strong scaling).
    scalars: ints or uint8 [n, 32] on the host, split evenly; or d_slices = [(e0_r, device pointer, n_r)] per replica for slices
    already resident on the replicas' GPUs (scalars is then ignored)."""
    replicas = list(replicas)
    arr = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
    out = C.create_string_buffer(96)
    if d_slices is not None:
        assert len(d_slices) == len(replicas)
        e = (C.c_int64 * len(replicas))(*[int(t[0]) for t in d_slices])
        ptrs = (C.c_void_p * len(replicas))(*[t[1].value if isinstance(t[1], C.c_void_p) else int(t[1]) for t in d_slices])
        cnt = (C.c_int64 * len(replicas))(*[int(t[2]) for t in d_slices])
        _lib.check(_lib.lib().sonic_msm_g1_srs_multi_dev(arr, len(replicas), basis, e, ptrs, cnt, mode, out))
        return out.raw
    sc = fr_array(scalars)
    _lib.check(_lib.lib().sonic_msm_g1_srs_multi(arr, len(replicas), basis, e0, sc.ctypes.data, sc.shape[0], mode, out))
    return out.raw


class MsmLane:
    """One MSM at a time over an SRS slice with device-resident scalars, in two halves (sonic_msm_submit / sonic_msm_collect).
    Two lanes used in turn stream MSMs: the sort and the reduction of one run under the accumulation of the other."""

    def __init__(self, device: int = -1):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().sonic_msm_lane_new_on(device, C.byref(self._h)))

    def submit(self, srs: SRS, basis: int, e0: int, d_scalars, n: int) -> None:
        """d_scalars: device pointer (int / c_void_p) to n canonical 32-byte Fr; must stay untouched until collect()"""
        _lib.check(_lib.lib().sonic_msm_submit(self._h, srs._h, basis, e0, d_scalars, n))

    def collect(self, partial: bool = False) -> bytes:
        """the 96 canonical bytes of the sum, or (partial=True) the 192-byte un-normalised partial for a cross-rank sum"""
        if partial:
            out = C.create_string_buffer(192)
            _lib.check(_lib.lib().sonic_msm_collect(self._h, None, out))
        else:
            out = C.create_string_buffer(96)
            _lib.check(_lib.lib().sonic_msm_collect(self._h, out, None))
        return out.raw

    def close(self):
        if self._h:
            _lib.lib().sonic_msm_lane_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
