#!/usr/bin/env python3
"""How the CPU oracle's Pippenger scales with threads on this host (the cpu_baseline of bench.py): one N-term MSM over an
oracle-made SRS per thread count, and what the host offers (visible CPUs, affinity mask, cgroup quota).
    python tools/cpu_scaling.py [--log2 18]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from sonic_amd.workload import rand_fr_array  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=int, default=18)
a = ap.parse_args()
print("os.cpu_count():", os.cpu_count(), " affinity:", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup cpu.max: n/a", e)
try:
    import subprocess
    print(subprocess.run("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
except Exception:
    pass
N = 1 << a.log2
d = N // 2
t0 = time.perf_counter()
srs = orc.SRS(d, 0x1234567, 0x7654321, threads=len(os.sched_getaffinity(0)))
print(f"orc.SRS(d=2^{a.log2 - 1}) {time.perf_counter() - t0:.2f}s with {len(os.sched_getaffinity(0))} threads")
sc = rand_fr_array(np.random.default_rng(1), N)
ref = None
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    if th > 2 * (os.cpu_count() or 1):
        break
    t0 = time.perf_counter()
    r = orc.msm_srs(srs, 0, -d, sc, 1, th)
    dt = time.perf_counter() - t0
    ref = ref or r
    assert r == ref
    print(f"threads={th:4d}  N=2^{a.log2}  {dt:8.3f}s  {N / dt:12.0f} scalar-muls/s  {N / dt / th:10.0f} per thread", flush=True)
