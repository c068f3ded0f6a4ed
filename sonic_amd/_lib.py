"""ctypes loader for sonic_amd/csrc/libsonic_hip.so (the C ABI of include/sonic_hip.h).

There is no fallback of any kind: if the HIP extension has not been built, importing the
product API fails with an ImportError that says how to build it; if no GPU is present every
entry point returns SONIC_ERR_NO_DEVICE and the Python layer raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import warnings

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SONIC_HIP_LIB") or os.path.join(_HERE, "csrc", "libsonic_hip.so")   # SONIC_HIP_LIB: another build of the same library (tuning experiments)

ERR_NAMES = {
    1: "D_TOO_SMALL", 2: "SRS_INDEX_OUT_OF_RANGE", 3: "BAD_ENCODING", 4: "INEXACT_DIVISION",
    5: "HIP_ERROR", 6: "NO_DEVICE", 7: "INVALID_ARG",
}


class SonicError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C sonic_amd/csrc). "
            "sonic_amd has no CPU fallback.")
    # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64; loaded AFTER this library
    # has pulled in /opt/rocm's, torch finds "No HIP GPUs" (two HSA runtimes cannot share the device).  Loaded first, torch's copy
    # also serves this library (same SONAME).  So: if torch is already imported, nothing to do; if it is installed but not yet
    # imported, it is imported here first -- unless SONIC_TORCH_PRELOAD=0 (or the older SONIC_NO_TORCH_PRELOAD) says the process
    # will never use torch (the C harness and a Haskell host never see any of this).  What got mapped is checked below.
    preload = os.environ.get("SONIC_TORCH_PRELOAD", "1") != "0" and os.environ.get("SONIC_NO_TORCH_PRELOAD") is None
    preloaded = None
    if preload and "torch" not in sys.modules:
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            try:
                import torch  # noqa: F401
                preloaded = True
            except Exception as e:      # noqa: BLE001
                warnings.warn(f"sonic_amd: importing torch before libsonic_hip.so failed ({e!r}); torch imported later in this process "
                              "will not find the GPU")
                preloaded = False
    L = C.CDLL(LIB_PATH)
    vp, cp, i64, i32 = C.c_void_p, C.c_char_p, C.c_int64, C.c_int
    sig = {
        "sonic_init": [i32],
        "sonic_last_error": [cp, C.c_size_t],
        "sonic_device_sync": [],
        "sonic_device_count": [C.POINTER(i32)],
        "sonic_srs_new": [i64, cp, cp, C.POINTER(vp)],
        "sonic_srs_new_on": [i32, i64, cp, cp, C.POINTER(vp)],
        "sonic_srs_from_points": [i64, vp, vp, C.POINTER(vp)],
        "sonic_srs_from_points_on": [i32, i64, vp, vp, C.POINTER(vp)],
        "sonic_srs_replicate": [vp, i32, C.POINTER(vp)],
        "sonic_srs_device": [vp],
        "sonic_srs_pairing": [vp, vp],
        "sonic_srs_load_on": [i32, cp, C.POINTER(vp)],
        "sonic_msm_lane_new_on": [i32, C.POINTER(vp)],
        "sonic_prove_shared": [vp, i32, vp, vp],
        "sonic_prove_batch": [vp, i32, i64, vp, vp, vp, vp, vp, vp],
        "sonic_prove_many": [vp, i32, i64, i64, vp, i64, vp, vp],
        "sonic_one_shot_trim": [i32],
        "sonic_msm_g1_srs_multi": [vp, i32, i32, i64, vp, i64, i32, vp],
        "sonic_msm_g1_srs_multi_dev": [vp, i32, i32, vp, vp, vp, i32, vp],
        "sonic_prover_device": [vp],
        "sonic_dev_alloc_on": [i32, C.c_size_t, C.POINTER(vp)],
        "sonic_srs_get_points": [vp, i32, i64, i64, vp],
        "sonic_srs_get_g2_points": [vp, i32, i64, i64, vp],
        "sonic_srs_save": [vp, cp, i32],
        "sonic_srs_has_g2": [vp],
        "sonic_srs_set_g2_points": [vp, vp, vp],
        "sonic_srs_load": [cp, C.POINTER(vp)],
        "sonic_commit_poly": [vp, i64, i64, vp, vp, vp],
        "sonic_open_poly": [vp, cp, i64, vp, vp, vp, vp],
        "sonic_msm_g1": [vp, vp, i64, vp],
        "sonic_msm_g1_srs": [vp, i32, i64, vp, i64, vp],
        "sonic_msm_g1_srs_dev": [vp, i32, i64, vp, i64, vp],
        "sonic_msm_g1_srs_partial_dev": [vp, i32, i64, vp, i64, vp],
        "sonic_g1_sum_partials": [vp, i32, vp],
        "sonic_g1_sum_dev_partials": [vp, i32, vp],
        "sonic_msm_lane_new": [C.POINTER(vp)],
        "sonic_msm_submit": [vp, vp, i32, i64, vp, i64],
        "sonic_msm_collect": [vp, vp, vp],
        "sonic_msm_lane_new_on_stream": [vp, C.POINTER(vp)],
        "sonic_msm_submit_dev_v2": [vp, vp, i32, i64, vp, i64, vp, C.c_size_t],
        "sonic_msm_exchange_layout": [vp, i32, C.POINTER(i64), C.POINTER(i64)],
        "sonic_msm_accumulate_dev": [vp, vp, i32, i64, vp, i64, vp, i64],
        "sonic_msm_reduce_slices_dev_v2": [vp, vp, vp, i32, i64, i64, vp, C.c_size_t],
        "sonic_msm_lane_sync": [vp],
        "sonic_ntt_fr": [vp, i32, i32],
        "sonic_poly_mul_fr": [vp, i64, vp, i64, vp],
        "sonic_poly_mul_fr_dev": [vp, i64, vp, i64, vp],
        "sonic_msm_set_window": [i32],
        "sonic_srs_point_bytes": [],
        "sonic_msm_plan": [vp, i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
        "sonic_prove": [vp, i64, i64] + [vp] * 9,
        "sonic_prover_new": [vp, i64, i64, vp, vp, vp, vp, C.POINTER(vp)],
        "sonic_prover_set_assignment": [vp, vp, vp, vp],
        "sonic_prover_prove": [vp, vp, vp],
        "sonic_prover_submit": [vp, vp],
        "sonic_prover_collect": [vp, vp],
        "sonic_prover_prepare": [vp],
        "sonic_prover_set_share": [vp, i32, i32],
        "sonic_prover_prove_share": [vp, vp, vp],
        "sonic_prover_collect_share": [vp, vp],
        "sonic_proof_from_shares": [i64, i32, vp, vp, vp],
        "sonic_prove_share_plan": [i64, i64, i32, i32, i32, i64, i32, vp, C.POINTER(C.c_double)],
        "sonic_fs_circuit_digest": [i64, i64, vp, vp, vp, vp, vp],
        "sonic_prover_prove_fs": [vp, cp, cp, vp, vp],
        "sonic_fs_srs_id": [vp, vp],
        "sonic_fs_challenges_v2": [i64, i64, i64, cp, cp, vp, vp],
        "sonic_verify_fs": [vp, i64, i64, vp, vp, vp, vp, vp, C.POINTER(i32)],
        "sonic_prover_hsc_prove": [vp, i64, vp, cp, cp, vp],
        "sonic_hsc_prove_poly": [vp, i64, vp, vp, vp, i64, vp, cp, cp, vp],
        "sonic_hsc_verify_poly": [vp, i64, vp, vp, vp, i64, vp, vp, C.POINTER(i32)],
        "sonic_hsc_verify": [vp, i64, i64, vp, vp, vp, i64, vp, vp, C.POINTER(i32)],
        "sonic_pc_v": [vp, i64, cp, cp, cp, cp, C.POINTER(i32)],
        "sonic_verify": [vp, i64, i64, vp, vp, vp, vp, vp, cp, cp, vp, C.POINTER(i32)],
        "sonic_dev_alloc": [C.c_size_t, C.POINTER(vp)],
        "sonic_dev_free": [vp],
        "sonic_dev_upload": [vp, vp, C.c_size_t],
        "sonic_dev_download": [vp, vp, C.c_size_t],
        "sonic_profile_enable": [i32],
        "sonic_profile_reset": [],
        "sonic_profile_get": [cp, C.POINTER(C.c_double), C.POINTER(i64)],
        "sonic_profile_names": [cp, C.c_size_t],
    }
    missing = [n for n in EXPORTED if not hasattr(L, n)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing}: stale build, rebuild it")
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    L.sonic_srs_free.argtypes = [vp]
    L.sonic_srs_free.restype = None
    L.sonic_prover_free.argtypes = [vp]
    L.sonic_prover_free.restype = None
    L.sonic_msm_lane_free.argtypes = [vp]
    L.sonic_msm_lane_free.restype = None
    L.sonic_srs_d.argtypes = [vp]
    L.sonic_srs_d.restype = i64
    L.sonic_proof_size.argtypes = [i64]
    L.sonic_proof_size.restype = C.c_size_t
    L.sonic_hsc_proof_size.argtypes = [i64]
    L.sonic_hsc_proof_size.restype = C.c_size_t
    L.sonic_proof_share_size.argtypes = [i64]
    L.sonic_proof_share_size.restype = C.c_size_t
    L.sonic_abi_version.argtypes = []
    L.sonic_abi_version.restype = i32
    if L.sonic_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {L.sonic_abi_version()}, this package binds version {ABI_VERSION}: rebuild it")
    # the retired symbols still link (and refuse): bound with their old prototypes so that tests can check exactly that
    L.sonic_msm_submit_dev.argtypes = [vp, vp, i32, i64, vp, i64, vp]
    L.sonic_msm_reduce_slices_dev.argtypes = [vp, vp, vp, i32, i64, i64, vp]
    L.sonic_fs_challenges.argtypes = [i64, i64, i64, cp, vp, vp]
    for fn in (L.sonic_msm_submit_dev, L.sonic_msm_reduce_slices_dev, L.sonic_fs_challenges):
        fn.restype = i32
    L.sonic_hip_versions.argtypes = [C.POINTER(i32), C.POINTER(i32)]
    L.sonic_hip_versions.restype = i32
    bv, rv = i32(), i32()
    L.sonic_hip_versions(C.byref(bv), C.byref(rv))
    if rv.value and bv.value // 10**5 != rv.value // 10**5:
        # not an error (the ABI this library uses is stable across these), but never silent
        msg = (f"sonic_amd: libsonic_hip.so was built against HIP {_hipver(bv.value)} and runs on the HIP runtime {_hipver(rv.value)} "
               f"that was mapped first{' (torch, imported by sonic_amd._lib)' if preloaded else ''}")
        if os.environ.get("SONIC_DEBUG"):
            print(msg, file=sys.stderr)
        global HIP_RUNTIME_NOTE
        HIP_RUNTIME_NOTE = msg
    _lib = L
    return L


HIP_RUNTIME_NOTE = None
ABI_VERSION = 6          # SONIC_ABI_VERSION of include/sonic_hip.h


def _hipver(v: int) -> str:
    return f"{v // 10**7}.{v // 10**5 % 100}.{v % 10**5}"


EXPORTED = [
    "sonic_abi_version", "sonic_device_count", "sonic_srs_new_on", "sonic_srs_from_points_on", "sonic_srs_replicate", "sonic_srs_device", "sonic_srs_pairing",
    "sonic_srs_load_on", "sonic_msm_lane_new_on", "sonic_msm_submit_dev_v2", "sonic_msm_reduce_slices_dev_v2", "sonic_fs_challenges_v2",
    "sonic_prove_shared", "sonic_prove_batch", "sonic_prove_many", "sonic_one_shot_trim", "sonic_msm_g1_srs_multi", "sonic_msm_g1_srs_multi_dev", "sonic_prover_device", "sonic_dev_alloc_on",
    "sonic_init", "sonic_last_error", "sonic_device_sync", "sonic_hip_versions", "sonic_srs_new", "sonic_srs_from_points",
    "sonic_srs_free", "sonic_srs_d", "sonic_srs_get_points", "sonic_srs_get_g2_points", "sonic_srs_set_g2_points", "sonic_srs_save", "sonic_srs_has_g2", "sonic_srs_load", "sonic_commit_poly", "sonic_open_poly",
    "sonic_msm_g1", "sonic_msm_g1_srs", "sonic_msm_g1_srs_dev", "sonic_msm_g1_srs_partial_dev",
    "sonic_g1_sum_partials", "sonic_g1_sum_dev_partials", "sonic_msm_lane_new", "sonic_msm_lane_free", "sonic_msm_submit", "sonic_msm_collect", "sonic_msm_lane_new_on_stream", "sonic_msm_submit_dev", "sonic_msm_exchange_layout",
    "sonic_msm_accumulate_dev", "sonic_msm_reduce_slices_dev", "sonic_msm_lane_sync", "sonic_ntt_fr", "sonic_poly_mul_fr", "sonic_poly_mul_fr_dev", "sonic_msm_set_window", "sonic_srs_point_bytes", "sonic_msm_plan",
    "sonic_proof_size", "sonic_prove", "sonic_prover_new", "sonic_prover_set_assignment",
    "sonic_prover_prove", "sonic_prover_submit", "sonic_prover_collect", "sonic_prover_prepare",
    "sonic_prover_set_share", "sonic_proof_share_size", "sonic_prover_prove_share", "sonic_prover_collect_share", "sonic_proof_from_shares", "sonic_prove_share_plan", "sonic_fs_circuit_digest", "sonic_fs_srs_id", "sonic_prover_prove_fs", "sonic_fs_challenges", "sonic_verify_fs", "sonic_prover_hsc_prove", "sonic_hsc_prove_poly", "sonic_hsc_verify_poly", "sonic_hsc_proof_size", "sonic_hsc_verify", "sonic_prover_free", "sonic_pc_v", "sonic_verify", "sonic_dev_alloc", "sonic_dev_free", "sonic_dev_upload",
    "sonic_dev_download", "sonic_profile_enable", "sonic_profile_reset", "sonic_profile_get",
    "sonic_profile_names",
]


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().sonic_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc: int) -> None:
    if rc != 0:
        raise SonicError(rc, last_error())
