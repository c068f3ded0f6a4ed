"""The generated gfx950 assembly on real hardware: tools/microbench --check runs, on the GPU, the Montgomery product, the
78-product squaring, the add / sub blocks (both representatives of the lazy range, 2.6e5 operand pairs) against the portable
C++ loops, and the fused mixed addition (six product, two squaring and one two-product core calls) against the general XYZZ
addition, exceptional lanes included.  tests/test_asm_model.py executes the same instruction streams on python integers."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_asm_routines_on_hardware():
    exe = os.path.join(ROOT, "tools", "microbench")
    src = os.path.join(ROOT, "tools", "microbench.hip")
    hdr = os.path.join(ROOT, "sonic_amd", "csrc", "mont_asm.hpp")
    hdr2 = os.path.join(ROOT, "sonic_amd", "csrc", "g1_quad.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(hdr2)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "sonic_amd", "csrc"),
                               "-I", os.path.join(ROOT, "include"), src, "-o", exe])
    out = subprocess.run([exe, "--check"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "quad addition vs general" in out.stdout and "asm fp_mul<Fq> vs C++ loop" in out.stdout and "FAIL" not in out.stdout and "self-checks ok" in out.stdout, out.stdout
