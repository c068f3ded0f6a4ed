#!/usr/bin/env python3
"""The protocol-shaped stand-alone MSMs of bench.py (`msm_protocol_shaped`: W_t's quotient and the coefficients of s(X,y), whose repeated
values make heavy buckets) alone, with the same number of uniform scalars beside each:  python tools/msm_shaped.py [--steps 10]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sonic_amd  # noqa: E402
from sonic_amd import _lib  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--which", default="both", choices=["both", "wt", "sy"], help="one of the two only (kernel traces)")
ap.add_argument("--no-uniform", action="store_true", help="skip the uniform-scalar comparison (kernel traces)")
a = ap.parse_args()
L = _lib.lib()
_lib.check(L.sonic_init(0))
x = 0x1234567890abcdef1234567890abcdef1234567890abcdef | 1
alpha = 0xfedcba0987654321fedcba0987654321fedcba09 | 1
print(json.dumps(bench.protocol_shaped_msm(sonic_amd, L, _lib, x, alpha, a.steps, a.warmup,
                                     which={"both": ("W_t_quotient", "s_of_X_y_coefficients"), "wt": ("W_t_quotient",), "sy": ("s_of_X_y_coefficients",)}[a.which],
                                     uniform=not a.no_uniform)))
