#!/usr/bin/env python3
"""Times the tPoly-shaped NTT product (M = 2^21 at n = 2^18) alone on the chip, per kernel (HIP events inside the library)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from sonic_amd import _lib  # noqa: E402
from util import rand_fr_array  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
    L = _lib.lib()
    _lib.check(L.sonic_init(0))
    na, nb = 3 * n + 5, 4 * n + 5
    pa, pb = rand_fr_array(np.random.default_rng(5), na), rand_fr_array(np.random.default_rng(6), nb)
    da, db, do = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for ptr, sz in ((da, 32 * na), (db, 32 * nb), (do, 32 * (na + nb - 1))):
        _lib.check(L.sonic_dev_alloc(sz, C.byref(ptr)))
    _lib.check(L.sonic_dev_upload(da, pa.ctypes.data, 32 * na))
    _lib.check(L.sonic_dev_upload(db, pb.ctypes.data, 32 * nb))
    for _ in range(3):
        _lib.check(L.sonic_poly_mul_fr_dev(da, na, db, nb, do))
    L.sonic_profile_reset()
    L.sonic_profile_enable(1)
    reps = 10
    for _ in range(reps):
        _lib.check(L.sonic_poly_mul_fr_dev(da, na, db, nb, do))
    L.sonic_profile_enable(0)
    per = {}
    for nm in ("k_ntt_wide", "k_ntt_local", "k_ntt_wide4", "k_ntt_local4", "k_ntt_wide_big"):
        ms, cnt = C.c_double(), C.c_int64()
        L.sonic_profile_get(nm.encode(), C.byref(ms), C.byref(cnt))
        per[nm] = round(ms.value / reps, 4)
    print("ntt product n =", n, "ms:", round(sum(per.values()), 4), per)


if __name__ == "__main__":
    main()
