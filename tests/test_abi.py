"""The C-ABI library loads and exports every symbol include/sonic_hip.h declares; without a GPU every
entry point refuses (no CPU fallback); the product package never imports the oracle."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sonic_amd", "csrc"), "-s", "-j8"])
    from sonic_amd import _lib
    return _lib


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "sonic_hip.h")).read()
    declared = set(re.findall(r"\b(sonic_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = C.CDLL(built.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(built.EXPORTED), declared ^ set(built.EXPORTED)


def _have_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_have_gpu(), reason="this box has a GPU; the refusal path needs none")
def test_no_device_is_loud(built):
    L = built.lib()
    assert L.sonic_init(0) == 6
    assert "no CPU fallback" in built.last_error()
    out = C.create_string_buffer(96)
    assert L.sonic_msm_g1(None, None, 0, out) == 6
    h = C.c_void_p()
    assert L.sonic_srs_new(16, (1).to_bytes(32, "little"), (2).to_bytes(32, "little"), C.byref(h)) == 6
    import sonic_amd
    with pytest.raises(sonic_amd.SonicError) as e:
        sonic_amd.SRS.new(16, 3, 5)
    assert e.value.code == 6


def test_missing_extension_is_loud(monkeypatch, built):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", os.path.join(ROOT, "sonic_amd", "csrc", "does_not_exist.so"))
    with pytest.raises(ImportError):
        built.lib()


def test_product_never_touches_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sonic_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                for pat in (r"import\s+oracle", r"from\s+oracle", r"oracle/", r"libsonic_oracle", r"sonic_ref", r"\borc\."):
                    assert not re.search(pat, src), (f, pat)


def _build_harness(tmp_path, name="abi_harness"):
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", name + ".c"), "-L" + os.path.join(ROOT, "sonic_amd", "csrc"), "-lsonic_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "sonic_amd", "csrc"), "-o", exe])
    return exe


@pytest.mark.skipif(_have_gpu(), reason="this box has a GPU: tests/test_gpu_parity.py runs the harness to the end")
def test_c99_harness_builds_and_refuses_without_gpu(built, tmp_path):
    """the boundary from plain C99 (the compile-checked stand-in for the Haskell `foreign import ccall` shim of INTEGRATION.md):
    the header is C99-clean, every entry point the harness binds links, and without a GPU the first call says so (exit 77)"""
    out = subprocess.run([_build_harness(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 77 and "SONIC_ERR_NO_DEVICE" in out.stderr and "no CPU fallback" in out.stderr


@pytest.mark.skipif(_have_gpu(), reason="this box has a GPU: tests/test_gpu_multi.py runs the harness to the end")
def test_c99_multi_harness_builds_and_refuses_without_gpu(built, tmp_path):
    """the multi-device entry points (sonic_srs_new_on, sonic_prove_shared, sonic_prove_batch, sonic_msm_g1_srs_multi ...) bind from
    plain C99; without a GPU the first call refuses"""
    case = tmp_path / "none.bin"
    case.write_bytes(b"")
    out = subprocess.run([_build_harness(tmp_path, "multi_harness"), str(case), "0,0,0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 77 and "SONIC_ERR_NO_DEVICE" in out.stderr


def test_abi_version_and_retired_symbols(built):
    """an entry point whose meaning changed got a new symbol; the old ones still link with their old prototypes and refuse
    (no device needed for that), and the library reports the ABI version the header declares"""
    import ctypes as C
    import re as _re
    L = built.lib()
    hdr = open(os.path.join(ROOT, "include", "sonic_hip.h")).read()
    assert L.sonic_abi_version() == int(_re.search(r"#define SONIC_ABI_VERSION (\d+)", hdr).group(1)) == built.ABI_VERSION
    out = C.create_string_buffer(96)
    assert L.sonic_fs_challenges(1, 1, 8, bytes(32), out, out) == 7 and "retired" in built.last_error()
    assert L.sonic_msm_submit_dev(None, None, 0, 0, None, 0, None) == 7 and "sonic_msm_submit_dev_v2" in built.last_error()
    assert L.sonic_msm_reduce_slices_dev(None, None, None, 1, 16384, 0, None) == 7 and "retired" in built.last_error()
