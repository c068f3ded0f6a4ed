"""Sonic.Protocol.prove with Sonic.Signature.hscProve (src/Sonic/Protocol.hs:47-109,
src/Sonic/Signature.hs:38-72) over the C ABI."""
from __future__ import annotations

import ctypes as C
import secrets
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .encoding import R_MODULUS, fr_array, fr_matrix, fr_to_bytes, g1_from_bytes, g1_to_bytes
from .srs import SRS


@dataclass
class GateWeights:           # Bulletproofs.ArithmeticCircuit.GateWeights
    wL: list
    wR: list
    wO: list


@dataclass
class ArithCircuit:          # Bulletproofs.ArithmeticCircuit.ArithCircuit (commitmentWeights is never forced)
    weights: GateWeights
    cs: list
    commitmentWeights: object = None


@dataclass
class Assignment:            # Bulletproofs.ArithmeticCircuit.Assignment
    aL: list
    aR: list
    aO: list


@dataclass
class HscProof:              # Signature.hs:22-29
    hscS: List[Tuple[object, Tuple[int, object]]]
    hscW: List[Tuple[int, object, object]]
    hscQv: object
    hscC: object
    hscU: int
    hscV: int


def _hsc_to_bytes(h: "HscProof") -> bytes:
    if len(h.hscS) != len(h.hscW):
        raise ValueError("HscProof: hscS and hscW must have one entry per (y_j, z_j) pair")

    def fr(v):                   # as is: a non-canonical field element must reach the verifier (which rejects it) unreduced
        return int(v).to_bytes(32, "little")
    out = []
    for cm, (sj, wj) in h.hscS:
        out += [g1_to_bytes(cm), fr(sj), g1_to_bytes(wj)]
    for sjp, wjp, qj in h.hscW:
        out += [fr(sjp), g1_to_bytes(wjp), g1_to_bytes(qj)]
    out += [g1_to_bytes(h.hscQv), g1_to_bytes(h.hscC), fr(h.hscU), fr(h.hscV)]
    return b"".join(out)


def _hsc_from_bytes(b: bytes, m: int) -> "HscProof":
    if len(b) != (2 + 4 * m) * 96 + (2 + 2 * m) * 32:
        raise ValueError(f"HscProof for m = {m} is {(2 + 4 * m) * 96 + (2 + 2 * m) * 32} bytes, got {len(b)}")
    pos = 0

    def g():
        nonlocal pos
        v = g1_from_bytes(b[pos:pos + 96]); pos += 96
        return v

    def f():
        nonlocal pos
        v = int.from_bytes(b[pos:pos + 32], "little"); pos += 32
        return v
    hscS = []
    for _ in range(m):
        cm, sj, wj = g(), f(), g()
        hscS.append((cm, (sj, wj)))
    hscW = []
    for _ in range(m):
        sjp, wjp, qj = f(), g(), g()
        hscW.append((sjp, wjp, qj))
    qv, c, u, v = g(), g(), f(), f()
    return HscProof(hscS, hscW, qv, c, u, v)


@dataclass
class Proof:                 # Protocol.hs:28-38
    prR: object
    prT: object
    prA: int
    prWa: object
    prB: int
    prWb: object
    prWt: object
    prS: int
    prHscProof: HscProof

    def to_bytes(self) -> bytes:
        """canonical proof bytes (include/sonic_hip.h): the record order of `Proof` then `HscProof`, serialised from the
        fields -- an edited or hand-built Proof is what gets verified, never a cached copy of the prover's output"""
        def fr(v):               # as is: a non-canonical field element must reach the verifier (which rejects it) unreduced
            return int(v).to_bytes(32, "little")
        head = [g1_to_bytes(self.prR), g1_to_bytes(self.prT), fr(self.prA), g1_to_bytes(self.prWa), fr(self.prB),
                g1_to_bytes(self.prWb), g1_to_bytes(self.prWt), fr(self.prS)]
        return b"".join(head) + _hsc_to_bytes(self.prHscProof)

    @classmethod
    def from_bytes(cls, b: bytes, Q: int) -> "Proof":
        b = bytes(b)
        if len(b) != (7 + 4 * Q) * 96 + (5 + 2 * Q) * 32:
            raise ValueError(f"proof for Q = {Q} is {(7 + 4 * Q) * 96 + (5 + 2 * Q) * 32} bytes, got {len(b)}")
        g = lambda o: g1_from_bytes(b[o:o + 96])                  # noqa: E731
        f = lambda o: int.from_bytes(b[o:o + 32], "little")       # noqa: E731
        # R T a Wa b Wb Wt s
        return cls(g(0), g(96), f(192), g(224), f(320), g(352), g(448), f(544), _hsc_from_bytes(b[576:], Q))


@dataclass
class RndOracle:             # Protocol.hs:40-45
    rndOracleY: int
    rndOracleZ: int
    rndOracleYZs: List[Tuple[int, int]]


def transcript_len(Q: int) -> int:
    return 8 + 2 * Q


def draw_transcript(Q: int, rng=None) -> List[int]:
    """The prover's `rnd` draws in draw order: c_{n+1..n+4}, y, z, ys, zs, then hscProve's u, v."""
    if rng is None:
        return [secrets.randbelow(R_MODULUS) for _ in range(transcript_len(Q))]
    return [rng.randrange(R_MODULUS) for _ in range(transcript_len(Q))]


class Prover:
    """Circuit (and assignment) resident in HBM across proofs: sonic_prover_* of the C ABI."""

    def __init__(self, srs: SRS, circuit: ArithCircuit, prepare: bool = True):
        w = circuit.weights
        wL, wR, wO = fr_matrix(w.wL), fr_matrix(w.wR), fr_matrix(w.wO)
        cs = fr_array(circuit.cs)
        self.Q = cs.shape[0]
        if self.Q < 1 or wL.shape[0] % self.Q:
            raise ValueError("need Q >= 1 rectangular weight rows")
        self.n = wL.shape[0] // self.Q
        assert wR.shape == wL.shape and wO.shape == wL.shape
        self._srs = srs
        self._h = C.c_void_p()
        _lib.check(_lib.lib().sonic_prover_new(srs._h, self.n, self.Q, wL.ctypes.data, wR.ctypes.data, wO.ctypes.data,
                                               cs.ctypes.data, C.byref(self._h)))
        if prepare:     # a handle exists to prove repeatedly: commit the constraint rows once (sonic_prover_prepare)
            _lib.check(_lib.lib().sonic_prover_prepare(self._h))

    def set_assignment(self, assignment: Assignment):
        aL, aR, aO = fr_array(assignment.aL), fr_array(assignment.aR), fr_array(assignment.aO)
        assert aL.shape[0] == self.n and aR.shape[0] == self.n and aO.shape[0] == self.n
        _lib.check(_lib.lib().sonic_prover_set_assignment(self._h, aL.ctypes.data, aR.ctypes.data, aO.ctypes.data))

    def prove_bytes(self, transcript) -> bytes:
        tr = fr_array(transcript)
        assert tr.shape[0] == transcript_len(self.Q)
        out = C.create_string_buffer(_lib.lib().sonic_proof_size(self.Q))
        _lib.check(_lib.lib().sonic_prover_prove(self._h, tr.ctypes.data, out))
        return out.raw

    def submit(self, transcript) -> None:
        """queue one proof on the GPU and return without waiting (sonic_prover_submit); one proof in flight per handle"""
        tr = fr_array(transcript)
        assert tr.shape[0] == transcript_len(self.Q)
        _lib.check(_lib.lib().sonic_prover_submit(self._h, tr.ctypes.data))

    def collect(self) -> bytes:
        """wait for the submitted proof, finish it on the host, return its bytes (sonic_prover_collect)"""
        out = C.create_string_buffer(_lib.lib().sonic_proof_size(self.Q))
        _lib.check(_lib.lib().sonic_prover_collect(self._h, out))
        return out.raw

    # ---- ONE proof over several GPUs (sonic_prover_set_share): this handle runs rank's pieces of the proof's MSMs ----
    def set_share(self, rank: int, world: int) -> None:
        _lib.check(_lib.lib().sonic_prover_set_share(self._h, rank, world))

    def prove_share(self, transcript) -> bytes:
        tr = fr_array(transcript)
        assert tr.shape[0] == transcript_len(self.Q)
        out = C.create_string_buffer(_lib.lib().sonic_proof_share_size(self.Q))
        _lib.check(_lib.lib().sonic_prover_prove_share(self._h, tr.ctypes.data, out))
        return out.raw

    def collect_share(self) -> bytes:
        """wait for the submitted share (submit() queues it like a whole proof) and return its bytes"""
        out = C.create_string_buffer(_lib.lib().sonic_proof_share_size(self.Q))
        _lib.check(_lib.lib().sonic_prover_collect_share(self._h, out))
        return out.raw

    def prove_fs(self, circuit_digest: bytes, blinder_seed: bytes):
        """prove with the opt-in Fiat-Shamir transcript (sonic_prover_prove_fs): returns (proof bytes, the 8 + 2Q transcript values
        the proof was made with); six waits for the GPU instead of one"""
        circuit_digest, blinder_seed = bytes(circuit_digest), bytes(blinder_seed)
        if len(circuit_digest) != 32 or len(blinder_seed) != 32:      # the C side reads 32 bytes of each
            raise ValueError("prove_fs: circuit_digest and blinder_seed must be 32 bytes each")
        out = C.create_string_buffer(_lib.lib().sonic_proof_size(self.Q))
        tr = C.create_string_buffer(32 * transcript_len(self.Q))
        _lib.check(_lib.lib().sonic_prover_prove_fs(self._h, bytes(circuit_digest), bytes(blinder_seed), out, tr))
        return out.raw, [int.from_bytes(tr.raw[32 * i:32 * i + 32], "little") for i in range(transcript_len(self.Q))]

    def hsc_prove(self, yzs, u: int, v: int) -> HscProof:
        """hscProve srs sXY yzs (Signature.hs:32-72) for the s(X,Y) of this handle's circuit; u, v: its two `rnd` draws"""
        yzs = list(yzs)
        flat = fr_array([x for pair in yzs for x in pair])
        out = C.create_string_buffer(_lib.lib().sonic_hsc_proof_size(len(yzs)))
        _lib.check(_lib.lib().sonic_prover_hsc_prove(self._h, len(yzs), flat.ctypes.data, fr_to_bytes(u), fr_to_bytes(v), out))
        return _hsc_from_bytes(out.raw, len(yzs))

    def close(self):
        if self._h:
            _lib.lib().sonic_prover_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def proof_from_shares(Q: int, shares, transcript) -> bytes:
    """the shares of all ranks (any order) -> canonical proof bytes (sonic_proof_from_shares; host only)"""
    shares = list(shares)
    blob = b"".join(bytes(s) for s in shares)
    tr = fr_array(transcript)
    out = C.create_string_buffer(_lib.lib().sonic_proof_size(Q))
    _lib.check(_lib.lib().sonic_proof_from_shares(Q, len(shares), blob, tr.ctypes.data, out))
    return out.raw


def share_plan(n: int, Q: int, prepared: bool, world: int, nb: int = 0, w: int = 0):
    """the plan of one proof over `world` ranks: per rank (pieces, cost) with pieces = 7 + 4Q pairs (lo, hi) in units of 1 / 2^20 of
    each MSM's terms (slot order R, T, W_a, W_b, W_t, [S_j, W_j]_j, [W'_j, Q_j]_j, Q_v, C)"""
    K = 7 + 4 * Q
    out = []
    for r in range(world):
        buf = (C.c_uint32 * (2 * K))()
        cost = C.c_double()
        _lib.check(_lib.lib().sonic_prove_share_plan(n, Q, int(prepared), world, r, nb, w, buf, C.byref(cost)))
        out.append(([(buf[2 * i], buf[2 * i + 1]) for i in range(K)], cost.value))
    return out


def device_count() -> int:
    """GPUs this process can drive (sonic_device_count)"""
    n = C.c_int(0)
    _lib.check(_lib.lib().sonic_device_count(C.byref(n)))
    return n.value


def _handle_array(provers):
    arr = (C.c_void_p * len(provers))(*[p._h for p in provers])
    return arr


def prove_shared(provers, transcript) -> bytes:
    """ONE proof made by several prover handles of the same circuit and assignment, one per GPU (or several on one GPU), from this
    one process: sonic_prove_shared runs handle r as rank r of len(provers) on a host thread of its own and combines the shares on
    the host.  Byte-identical to Prover.prove_bytes on one GPU.  No torch, no RCCL (the one-process-per-GPU form is
    sonic_amd.distributed.ShardedProver)."""
    provers = list(provers)
    Q = provers[0].Q
    tr = fr_array(transcript)
    assert tr.shape[0] == transcript_len(Q)
    out = C.create_string_buffer(_lib.lib().sonic_proof_size(Q))
    _lib.check(_lib.lib().sonic_prove_shared(_handle_array(provers), len(provers), tr.ctypes.data, out))
    return out.raw


def prove_batch(provers, transcripts, assignments=None) -> List[bytes]:
    """`mapM prove` over K statements of one circuit, spread over several prover handles (sonic_prove_batch: proof i on handle
    i % len(provers), one host thread per handle, no collective): the throughput mode of BASELINE's "batch of 64 independent proofs
    streamed over 8 GPUs".  assignments: K Assignment objects, or None to prove every statement with the handles' resident
    assignment (then only the transcripts differ)."""
    provers = list(provers)
    Q, n = provers[0].Q, provers[0].n
    K = len(transcripts)
    tr = np.ascontiguousarray(np.stack([fr_array(t) for t in transcripts]) if K else np.zeros((0, transcript_len(Q), 32), np.uint8))
    assert tr.shape[1] == transcript_len(Q)
    psz = _lib.lib().sonic_proof_size(Q)
    out = np.zeros((max(K, 1), psz), np.uint8)
    status = (C.c_int * max(K, 1))()
    aL = aR = aO = None
    if assignments is not None:
        assert len(assignments) == K
        aL = np.ascontiguousarray(np.stack([fr_array(a.aL) for a in assignments]))
        aR = np.ascontiguousarray(np.stack([fr_array(a.aR) for a in assignments]))
        aO = np.ascontiguousarray(np.stack([fr_array(a.aO) for a in assignments]))
        assert aL.shape[1] == n and aR.shape == aL.shape and aO.shape == aL.shape
    ptr = lambda a: None if a is None else a.ctypes.data       # noqa: E731
    _lib.check(_lib.lib().sonic_prove_batch(_handle_array(provers), len(provers), K, ptr(aL), ptr(aR), ptr(aO), tr.ctypes.data, out.ctypes.data, status))
    return [out[i].tobytes() for i in range(K)]


class _Statement(C.Structure):          # sonic_statement_t
    _fields_ = [(k, C.c_void_p) for k in ("wL", "wR", "wO", "cs", "aL", "aR", "aO", "transcript")]


def prove_many(replicas, statements) -> List[bytes]:
    """`mapM (\\(asg, circ, tr) -> prove srs asg circ)` over K INDEPENDENT statements of one shape (n, Q) -- every proof its own circuit,
    assignment and transcript -- spread over the SRS replicas, one per GPU (sonic_prove_many: two host threads per replica making
    one-shot calls; no collective).  statements: (Assignment, ArithCircuit, transcript) triples.  BASELINE's "batch of 64 independent
    proofs streamed over 8 GPUs"."""
    replicas, statements = list(replicas), list(statements)
    K = len(statements)
    if K == 0:
        return []
    keep, arr = [], (_Statement * K)()
    n = Q = None
    for i, (asg, circ, tr) in enumerate(statements):
        wL, wR, wO, cs, n_i, Q_i = _circuit_arrays(circ)
        aL, aR, aO, t = fr_array(asg.aL), fr_array(asg.aR), fr_array(asg.aO), fr_array(tr)
        if n is None:
            n, Q = n_i, Q_i
        if (n_i, Q_i) != (n, Q) or aL.shape[0] != n or aR.shape[0] != n or aO.shape[0] != n or t.shape[0] != transcript_len(Q):
            raise ValueError(f"prove_many: statement {i} has another shape than statement 0 (n = {n}, Q = {Q})")
        bufs = (wL, wR, wO, cs, aL, aR, aO, t)
        keep.append(bufs)
        for name, b in zip(("wL", "wR", "wO", "cs", "aL", "aR", "aO", "transcript"), bufs):
            setattr(arr[i], name, b.ctypes.data)
    psz = _lib.lib().sonic_proof_size(Q)
    out = np.zeros((K, psz), np.uint8)
    status = (C.c_int * K)()
    srs_arr = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
    _lib.check(_lib.lib().sonic_prove_many(srs_arr, len(replicas), n, Q, arr, K, out.ctypes.data, status))
    return [out[i].tobytes() for i in range(K)]


class ProverPipeline:
    """`mapM prove` over a stream of statements of one circuit, from one host thread: `depth` prover handles used in turn, so that
    while proof i is being waited for and finished, proof i + 1 is already running (its polynomial building and sorts fill the
    reduction tail of proof i).  Same bytes as proving one after the other."""

    def __init__(self, srs: SRS, circuit: ArithCircuit, depth: int = 2, prepare: bool = True):
        self.provers = [Prover(srs, circuit, prepare) for _ in range(max(1, depth))]

    def set_assignment(self, assignment: Assignment):
        for p in self.provers:
            p.set_assignment(assignment)

    def prove_all(self, transcripts) -> List[bytes]:
        k = len(self.provers)
        out: List[bytes] = []
        inflight: List[int] = []          # indices of the submitted, not yet collected proofs, oldest first
        try:
            for i, tr in enumerate(transcripts):
                if len(inflight) == k:
                    out.append(self.provers[inflight.pop(0) % k].collect())
                self.provers[i % k].submit(tr)
                inflight.append(i)
            while inflight:
                out.append(self.provers[inflight.pop(0) % k].collect())
        finally:
            for i in inflight:            # an error on the way: leave no handle with a proof in flight
                try:
                    self.provers[i % k].collect()
                except Exception:
                    pass
        return out

    def close(self):
        for p in self.provers:
            p.close()


def prove(srs: SRS, assignment: Assignment, circuit: ArithCircuit, transcript: Optional[list] = None, rng=None):
    """prove :: SRS -> Assignment Fr -> ArithCircuit Fr -> m (Proof, RndOracle) (Protocol.hs:47-52).
    `transcript` makes the MonadRandom draws explicit (reproducible proofs); default: fresh draws."""
    n = len(assignment.aL) if not isinstance(assignment.aL, np.ndarray) else fr_array(assignment.aL).shape[0]
    Q = fr_array(circuit.cs).shape[0]
    if srs.srsD < 7 * n:   # Protocol.hs:54-55 (checked again by the library)
        raise _lib.SonicError(1, f"Parameter d is not large enough: {srs.srsD} should be greater than {7 * n}")
    if transcript is None:
        transcript = draw_transcript(Q, rng)
    # the one-shot entry point (sonic_prove): circuit, assignment and transcript as host buffers, like the reference's call; the library
    # parks the handle's shell (streams, workspaces, twiddle tables) for the next call of the same shape -- making and freeing a handle
    # per call cost ~7 ms of stream / pinned-memory set-up around a 2-ms proof at the reference's own benchmark sizes
    wL, wR, wO, cs, n_c, Q_c = _circuit_arrays(circuit)
    aL, aR, aO = fr_array(assignment.aL), fr_array(assignment.aR), fr_array(assignment.aO)
    if n_c != n or aL.shape[0] != n or aR.shape[0] != n or aO.shape[0] != n:
        raise ValueError("assignment and weight rows differ in length")
    tr = fr_array(transcript)
    if tr.shape[0] != 8 + 2 * Q:
        raise ValueError(f"transcript needs 8 + 2Q = {8 + 2 * Q} elements")
    out = C.create_string_buffer(_lib.lib().sonic_proof_size(Q))
    _lib.check(_lib.lib().sonic_prove(srs._h, n, Q, wL.ctypes.data, wR.ctypes.data, wO.ctypes.data, cs.ctypes.data,
                                      aL.ctypes.data, aR.ctypes.data, aO.ctypes.data, tr.ctypes.data, out))
    raw = out.raw
    t = [int(v) % R_MODULUS for v in transcript]
    oracle = RndOracle(t[4], t[5], list(zip(t[6:6 + Q], t[6 + Q:6 + 2 * Q])))
    return Proof.from_bytes(raw, Q), oracle


def _circuit_arrays(circuit: ArithCircuit):
    w = circuit.weights
    wL, wR, wO = fr_matrix(w.wL), fr_matrix(w.wR), fr_matrix(w.wO)
    cs = fr_array(circuit.cs)
    Q = cs.shape[0]
    if Q < 1 or wL.shape[0] % Q or wL.shape[0] == 0 or wR.shape != wL.shape or wO.shape != wL.shape:
        raise ValueError("need Q >= 1 weight rows of equal length n >= 1 in wL, wR, wO")
    return wL, wR, wO, cs, wL.shape[0] // Q, Q


def fs_circuit_digest(circuit: ArithCircuit) -> bytes:
    """SHA-256 of (n, Q, wL, wR, wO, cs): the statement part of the Fiat-Shamir transcript, once per circuit"""
    wL, wR, wO, cs, n, Q = _circuit_arrays(circuit)
    out = C.create_string_buffer(32)
    _lib.check(_lib.lib().sonic_fs_circuit_digest(n, Q, wL.ctypes.data, wR.ctypes.data, wO.ctypes.data, cs.ctypes.data, out))
    return out.raw


def fs_srs_id(srs: SRS) -> bytes:
    """SHA-256 of d, g^x, g^{alpha x}, g^{1/x}, g^{alpha/x}: what binds a Fiat-Shamir transcript to one reference string"""
    out = C.create_string_buffer(32)
    _lib.check(_lib.lib().sonic_fs_srs_id(srs._h, out))
    return out.raw


def fs_challenges(srs: SRS, circuit: ArithCircuit, proof: Proof) -> RndOracle:
    """the RndOracle a Fiat-Shamir proof determines (sonic_fs_challenges_v2)"""
    wL, wR, wO, cs, n, Q = _circuit_arrays(circuit)
    raw = proof.to_bytes()
    if len(raw) != _lib.lib().sonic_proof_size(Q):
        raise ValueError("fs_challenges: the proof does not have Q entries in its hsc lists")
    out = C.create_string_buffer(32 * (4 + 2 * Q))
    _lib.check(_lib.lib().sonic_fs_challenges_v2(n, Q, srs.srsD, fs_circuit_digest(circuit), fs_srs_id(srs), raw, out))
    v = [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(4 + 2 * Q)]
    return RndOracle(v[0], v[1], list(zip(v[2:2 + Q], v[2 + Q:2 + 2 * Q])))


def prove_fs(srs: SRS, assignment: Assignment, circuit: ArithCircuit, blinder_seed: Optional[bytes] = None):
    """prove with every `rnd` draw of the reference (Protocol.hs:58,66,76,84-85; Signature.hs:48,60) replaced by the hash of what
    precedes it: (Proof, RndOracle); the blinders come from `blinder_seed` (32 bytes, default fresh)"""
    seed = secrets.token_bytes(32) if blinder_seed is None else bytes(blinder_seed)
    p = Prover(srs, circuit, prepare=False)
    try:
        p.set_assignment(assignment)
        raw, t = p.prove_fs(fs_circuit_digest(circuit), seed)
    finally:
        p.close()
    Q = p.Q
    return Proof.from_bytes(raw, Q), RndOracle(t[4], t[5], list(zip(t[6:6 + Q], t[6 + Q:6 + 2 * Q])))


def verify_fs(srs: SRS, circuit: ArithCircuit, proof: Proof) -> bool:
    """verify for a Fiat-Shamir proof: the challenges are recomputed from the circuit and the proof (sonic_verify_fs)"""
    wL, wR, wO, cs, n, Q = _circuit_arrays(circuit)
    h = proof.prHscProof
    if len(h.hscS) != Q or len(h.hscW) != Q:
        return False
    raw = proof.to_bytes()
    ok = C.c_int(0)
    _lib.check(_lib.lib().sonic_verify_fs(srs._h, n, Q, wL.ctypes.data, wR.ctypes.data, wO.ctypes.data, cs.ctypes.data, raw, C.byref(ok)))
    return bool(ok.value)


def hsc_prove(srs: SRS, circuit: ArithCircuit, yzs, u: Optional[int] = None, v: Optional[int] = None, rng=None) -> HscProof:
    """hscProve :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> m HscProof (Signature.hs:32-72), with sXY = sPoly of `circuit`'s weights
    (Constraints.hs:34-53); u, v default to fresh draws"""
    draw = (lambda: rng.randrange(1, R_MODULUS)) if rng is not None else (lambda: secrets.randbelow(R_MODULUS - 1) + 1)
    p = Prover(srs, circuit, prepare=False)
    try:
        return p.hsc_prove(yzs, draw() if u is None else u, draw() if v is None else v)
    finally:
        p.close()


def from_x(p) -> dict:
    """fromX :: VLaurent f -> BiVLaurent f (Utils.hs:23-24): the univariate polynomial {exponent: coeff} in X as a bivariate one,
    constant in Y -- in the nested form hsc_prove_poly takes.  The prover itself never lifts: evaluating Y := y is a ring
    homomorphism, so tPoly's product (Constraints.hs:56-65) is formed in the univariate ring."""
    return {int(ex): {0: int(c)} for ex, c in (p.items() if isinstance(p, dict) else p) if int(c) % R_MODULUS}


def from_y(p) -> dict:
    """fromY :: VLaurent f -> BiVLaurent f (Utils.hs:26-27): `monomial 0` -- a polynomial in Y as the X^0 coefficient"""
    inner = {int(ey): int(c) for ey, c in (p.items() if isinstance(p, dict) else p) if int(c) % R_MODULUS}
    return {0: inner} if inner else {}


def biv_add(a: dict, b: dict) -> dict:
    """sum of two BiVLaurent polynomials in the nested form (coefficients mod r, zero terms dropped)"""
    out = {ex: dict(inner) for ex, inner in a.items()}
    for ex, inner in b.items():
        row = out.setdefault(ex, {})
        for ey, c in inner.items():
            row[ey] = (row.get(ey, 0) + c) % R_MODULUS
    return {ex: {ey: c for ey, c in inner.items() if c} for ex, inner in out.items() if any(inner.values())}


def _biv_terms(sXY):
    """BiVLaurent Fr as {x_exp: {y_exp: coeff}} (X outside, Y inside, like poly's nested sparse form) or [(x_exp, y_exp, coeff)]"""
    if isinstance(sXY, dict):
        items = [(ex, ey, c) for ex, inner in sXY.items() for ey, c in inner.items()]
    else:
        items = [(ex, ey, c) for ex, ey, c in sXY]
    xe = np.array([t[0] for t in items], dtype=np.int64)
    ye = np.array([t[1] for t in items], dtype=np.int64)
    return xe, ye, fr_array([t[2] for t in items])


def hsc_prove_poly(srs: SRS, sXY, yzs, u: Optional[int] = None, v: Optional[int] = None, rng=None) -> HscProof:
    """hscProve :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> m HscProof (Signature.hs:32-72) for ANY sparse bivariate Laurent
    polynomial (sonic_hsc_prove_poly); u, v: its two `rnd` draws (default fresh)"""
    draw = (lambda: rng.randrange(1, R_MODULUS)) if rng is not None else (lambda: secrets.randbelow(R_MODULUS - 1) + 1)
    xe, ye, cf = _biv_terms(sXY)
    yzs = list(yzs)
    flat = fr_array([x for pair in yzs for x in pair])
    out = C.create_string_buffer(_lib.lib().sonic_hsc_proof_size(len(yzs)))
    _lib.check(_lib.lib().sonic_hsc_prove_poly(srs._h, len(xe), xe.ctypes.data, ye.ctypes.data, cf.ctypes.data, len(yzs), flat.ctypes.data,
                                               fr_to_bytes(draw() if u is None else u), fr_to_bytes(draw() if v is None else v), out))
    return _hsc_from_bytes(out.raw, len(yzs))


def hsc_verify_poly(srs: SRS, sXY, yzs, proof: HscProof) -> bool:
    """hscVerify :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> HscProof -> Bool (Signature.hs:74-90) for any sparse bivariate polynomial"""
    xe, ye, cf = _biv_terms(sXY)
    yzs = list(yzs)
    if len(proof.hscS) != len(yzs) or len(proof.hscW) != len(yzs):
        return False
    flat = fr_array([x for pair in yzs for x in pair])
    ok = C.c_int(0)
    _lib.check(_lib.lib().sonic_hsc_verify_poly(srs._h, len(xe), xe.ctypes.data, ye.ctypes.data, cf.ctypes.data, len(yzs), flat.ctypes.data,
                                                _hsc_to_bytes(proof), C.byref(ok)))
    return bool(ok.value)


def hsc_verify(srs: SRS, circuit: ArithCircuit, yzs, proof: HscProof) -> bool:
    """hscVerify :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> HscProof -> Bool (Signature.hs:74-90); host CPU pairings"""
    w = circuit.weights
    wL, wR, wO = fr_matrix(w.wL), fr_matrix(w.wR), fr_matrix(w.wO)
    Q = fr_array(circuit.cs).shape[0]
    if Q < 1 or wL.shape[0] % Q or wL.shape[0] == 0 or wR.shape != wL.shape or wO.shape != wL.shape:
        raise ValueError("hsc_verify: need Q >= 1 weight rows of equal length n >= 1 in wL, wR, wO")
    n = wL.shape[0] // Q
    yzs = list(yzs)
    if any(len(pair) != 2 for pair in yzs):
        raise ValueError("hsc_verify: yzs must hold (y_j, z_j) pairs")
    if len(proof.hscS) != len(yzs) or len(proof.hscW) != len(yzs):
        return False
    flat = fr_array([x for pair in yzs for x in pair])
    ok = C.c_int(0)
    _lib.check(_lib.lib().sonic_hsc_verify(srs._h, n, Q, wL.ctypes.data, wR.ctypes.data, wO.ctypes.data, len(yzs), flat.ctypes.data,
                                           _hsc_to_bytes(proof), C.byref(ok)))
    return bool(ok.value)


def verify(srs: SRS, circuit: ArithCircuit, proof: Proof, y: int, z: int, yzs) -> bool:
    """verify :: SRS -> ArithCircuit Fr -> Proof -> Fr -> Fr -> [(Fr, Fr)] -> Bool (Protocol.hs:111-130), with
    hscVerify (Signature.hs:74-90).  Runs on the host CPU (pairings); the SRS needs its G2 half (SRS.new, or a file
    that carries it).  Shapes are checked here because the C side reads 64 Q bytes of yzs and sonic_proof_size(Q)
    bytes of proof; a Proof whose hsc lists do not have Q entries is rejected (False), like any other malformed proof."""
    w = circuit.weights
    wL, wR, wO = fr_matrix(w.wL), fr_matrix(w.wR), fr_matrix(w.wO)
    cs = fr_array(circuit.cs)
    Q = cs.shape[0]
    if Q < 1 or wL.shape[0] % Q or wL.shape[0] == 0 or wR.shape != wL.shape or wO.shape != wL.shape:
        raise ValueError("verify: need Q >= 1 weight rows of equal length n >= 1 in wL, wR, wO")
    n = wL.shape[0] // Q
    yzs = list(yzs)
    if len(yzs) != Q or any(len(pair) != 2 for pair in yzs):
        raise ValueError(f"verify: yzs must hold {Q} (y_j, z_j) pairs, got {len(yzs)}")
    h = proof.prHscProof
    if len(h.hscS) != Q or len(h.hscW) != Q:
        return False
    raw = proof.to_bytes()
    if len(raw) != _lib.lib().sonic_proof_size(Q):
        return False
    flat = fr_array([v for pair in yzs for v in pair])
    ok = C.c_int(0)
    _lib.check(_lib.lib().sonic_verify(srs._h, n, Q, wL.ctypes.data, wR.ctypes.data, wO.ctypes.data, cs.ctypes.data, raw,
                                       fr_to_bytes(y), fr_to_bytes(z), flat.ctypes.data, C.byref(ok)))
    return bool(ok.value)
