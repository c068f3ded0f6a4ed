#!/usr/bin/env python3
"""Stand-alone MSMs over a SMALL SRS (one job of 2^15 / 2^16 buckets: the launches that four lanes per bucket are for), scalars resident
in HBM, one MSM at a time: ms per MSM.  `SONIC_ACCUM_LANES=1 python tools/msm_small.py` = one lane per bucket (round 5)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sonic_amd  # noqa: E402
from sonic_amd import _lib  # noqa: E402
from sonic_amd.workload import rand_fr_array  # noqa: E402

L = _lib.lib()
_lib.check(L.sonic_init(0))
rng = np.random.default_rng(1)
for lgd in (16, 17, 19):
    d = 1 << lgd
    srs = sonic_amd.SRS.new(d, 0x1234567, 0x7654321)
    pc, pw, pb = C.c_int(), C.c_int(), C.c_int()
    for lgn in (12, 14, 16, lgd):
        N = 1 << lgn
        _lib.check(L.sonic_msm_plan(srs._h, N, C.byref(pc), C.byref(pw), C.byref(pb)))
        sc = rand_fr_array(rng, N)
        dp = C.c_void_p()
        _lib.check(L.sonic_dev_alloc(32 * N, C.byref(dp)))
        _lib.check(L.sonic_dev_upload(dp, sc.ctypes.data, 32 * N))
        out = C.create_string_buffer(96)
        for _ in range(5):
            _lib.check(L.sonic_msm_g1_srs_dev(srs._h, 0, -N // 2, dp, N, out))
        L.sonic_device_sync()
        t0 = time.perf_counter()
        for _ in range(30):
            _lib.check(L.sonic_msm_g1_srs_dev(srs._h, 0, -N // 2, dp, N, out))
        dt = (time.perf_counter() - t0) / 30
        print(f"d=2^{lgd} (c={pc.value}, {pw.value} windows, {1 << (pc.value - 1)} buckets)  N=2^{lgn}: {dt * 1e3:.3f} ms per MSM  result {out.raw[:4].hex()}")
        L.sonic_dev_free(dp)
    srs.close()
