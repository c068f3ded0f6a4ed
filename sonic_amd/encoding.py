"""Canonical encodings at the C ABI (include/sonic_hip.h): Fr = 32 B LE, G1 = 96 B x||y LE,
infinity = 96 zero bytes.  Python-side values: Fr = int, G1 = (x, y) ints or None (`mempty`)."""
from __future__ import annotations

import numpy as np

R_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Q_MODULUS = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def fr_to_bytes(x: int) -> bytes:
    return (int(x) % R_MODULUS).to_bytes(32, "little")


def fr_from_bytes(b: bytes) -> int:
    return int.from_bytes(b, "little")


def g1_to_bytes(p) -> bytes:
    if p is None:
        return bytes(96)
    return int(p[0]).to_bytes(48, "little") + int(p[1]).to_bytes(48, "little")


def g1_from_bytes(b: bytes):
    b = bytes(b)
    if b == bytes(96):
        return None
    return (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:], "little"))


def fr_array(vals) -> np.ndarray:
    """ints, or an already-encoded uint8 array [k, 32] -> contiguous uint8 [k, 32]."""
    if isinstance(vals, np.ndarray):
        a = np.ascontiguousarray(vals, dtype=np.uint8)
        return a.reshape(-1, 32)
    vals = list(vals)
    if not vals:
        return np.zeros((0, 32), np.uint8)
    return np.frombuffer(b"".join(fr_to_bytes(v) for v in vals), dtype=np.uint8).reshape(-1, 32).copy()


def fr_matrix(rows) -> np.ndarray:
    """list of Q rows of n ints (Bulletproofs GateWeights, `[[f]]`) or uint8 [Q, n, 32] -> uint8 [Q*n, 32]."""
    if isinstance(rows, np.ndarray):
        return np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, 32)
    return fr_array([v for row in rows for v in row])
