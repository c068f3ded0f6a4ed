"""sonic_amd -- MI355X-native Sonic prover hot path (drop-in for sdiehl/sonic's prove path).

Python host-side mirror of the reference's Haskell modules over the C ABI of
``sonic_amd/csrc/libsonic_hip.so`` (``include/sonic_hip.h``):

    Sonic.SRS               -> sonic_amd.srs          (SRS, SRS.new)
    Sonic.CommitmentScheme  -> sonic_amd.commitment   (commit_poly, open_poly, pc_v)
    Sonic.Protocol          -> sonic_amd.protocol     (prove, verify, Proof, RndOracle, Prover)
    Sonic.Signature         -> sonic_amd.protocol     (HscProof, hsc_prove, hsc_verify; hscProve also runs inside prove)

All compute happens in hand-written HIP kernels on the GPU; this package is ctypes plumbing.
There is no CPU fallback: without the built extension imports fail, without a GPU calls raise.
"""
import os as _os

# prove() drives the t(X,y) product and up to six MSM groups on their own HIP streams; the ROCm runtime multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one run behind each other.  Read when the HIP
# runtime initialises, so this only takes effect if the package is imported before the process first touches the GPU.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import SonicError, LIB_PATH  # noqa: F401,E402
from .encoding import R_MODULUS, Q_MODULUS, fr_to_bytes, fr_from_bytes, g1_to_bytes, g1_from_bytes  # noqa: F401,E402
from .srs import SRS  # noqa: F401,E402
from .commitment import commit_poly, open_poly, pc_v, msm_g1, MsmLane  # noqa: F401,E402
from .protocol import hsc_prove_poly, hsc_verify_poly  # noqa: F401,E402
from .protocol import prove_fs, verify_fs, fs_challenges, fs_circuit_digest, fs_srs_id  # noqa: F401,E402
from .protocol import proof_from_shares, share_plan, from_x, from_y, biv_add  # noqa: F401,E402
from .protocol import prove_shared, prove_batch, prove_many, device_count  # noqa: F401,E402
from .commitment import msm_g1_srs_multi  # noqa: F401,E402
from .protocol import prove, verify, hsc_prove, hsc_verify, Proof, HscProof, RndOracle, Prover, ProverPipeline, ArithCircuit, Assignment, GateWeights  # noqa: F401,E402

__all__ = ["SRS", "commit_poly", "open_poly", "pc_v", "msm_g1", "MsmLane", "prove", "verify", "prove_fs", "verify_fs", "fs_challenges", "fs_circuit_digest", "hsc_prove", "hsc_verify", "hsc_prove_poly", "hsc_verify_poly", "Proof", "HscProof", "RndOracle", "Prover", "ProverPipeline",
           "ArithCircuit", "Assignment", "GateWeights", "SonicError"]
