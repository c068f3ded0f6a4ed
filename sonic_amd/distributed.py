"""Multi-GPU plumbing for the one step of the path that shards inside a single MSM.

sum_i s_i * P_i (the folds at src/Sonic/CommitmentScheme.hs:25-29, 45-48) is a sum of independent terms; one process per GPU,
SRS replicated, and two ways of splitting ONE MSM over the ranks:

* by TERM range (`ShardedMsm.run_terms`, weak-scaling layout of bench.py): rank r runs the whole bucket method on its slice and
  contributes one un-normalised XYZZ partial (192 bytes); all-gather of the partials, k-1 curve additions + one normalisation on
  every rank.  No bulk traffic, but every rank reduces a full bucket set, which does not shrink with the shard.
* by BUCKET range (`ShardedMsm.run_buckets`, strong scaling): rank r accumulates its term slice into a full bucket set, the ranks
  exchange bucket ranges with ONE all-to-all (n_buckets x 192 B / world per pair: 12.6 MB at 2^19 buckets over 8 ranks), each
  rank adds the slices it received and reduces only its 1/world of the buckets; then the same 192-byte all-gather.

RCCL has no elliptic-curve reduction op, so both end in a gather + local curve additions, never an all-reduce.  The collectives
run on device tensors ordered on one HIP stream with the lane's kernels (sonic_msm_lane_new_on_stream): nothing visits the host
between the scalars in HBM and the gathered partials, which come back in ONE device-to-host copy.  With the gloo backend (tests:
several ranks on one GPU, or no GPU at all) the same exchanges are staged through host tensors.

prove() shards two ways: by proof (every rank its own proofs, no collective: bench.py's `value`), and -- `ShardedProver` -- ONE
proof over all ranks: the 7 + 4Q commitments and openings of prove + hscProve (src/Sonic/Protocol.hs:63,73,79-81,
src/Sonic/Signature.hs:40-45,51-57,63) are independent sums once the transcript is known, so every rank builds the polynomials
its pieces read, runs a contiguous, cost-balanced piece of those MSMs (sonic_prover_set_share) and the ranks all-gather their
shares (a few KB: 192-byte partial sums, evaluations, error flags).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib

PARTIAL_BYTES = 192
DEV_PARTIAL_BYTES = 12304       # SONIC_G1_DEV_PARTIAL_BYTES: a device-side MSM result (header + up to 64 points the host folds)


def dev_partial_from_sum(partial192: bytes) -> bytes:
    """a finished 192-byte XYZZ sum in the device-partial form: one window sum (W = 1, c = 1)"""
    import struct
    return struct.pack("<iiii", 1, 1, 0, 0) + bytes(partial192) + bytes(DEV_PARTIAL_BYTES - 16 - PARTIAL_BYTES)


def sum_dev_partials(blobs: np.ndarray, k: int) -> bytes:
    """the sum of k device-side MSM results (DEV_PARTIAL_BYTES each), normalised -> 96 canonical bytes"""
    p = np.ascontiguousarray(blobs, np.uint8)
    out = C.create_string_buffer(96)
    _lib.check(_lib.lib().sonic_g1_sum_dev_partials(p.ctypes.data, k, out))
    return out.raw


def msm_shard(rank: int, world: int, d: int, terms_per_rank: int) -> Tuple[int, int]:
    """(basis, first exponent) of rank's slice: consecutive slices of basis 0 starting at -d, wrapping
    into basis 1 when 2d/terms_per_rank slices are used up (weak-scaling benchmark layout)."""
    per_basis = max(1, (2 * d) // terms_per_rank)
    if rank >= 2 * per_basis:
        raise ValueError(f"SRS of degree {d} holds {2 * per_basis} slices of {terms_per_rank} terms; rank {rank} does not fit")
    return rank // per_basis, -d + (rank % per_basis) * terms_per_rank


def split_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """strong-scaling split of one n-term MSM: contiguous [lo, hi) for rank (last ranks may be empty)"""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def _pg_active() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def allgather_partials(partial: np.ndarray, world: int, device=None) -> np.ndarray:
    """all-gather of HOST 192-byte partials over torch.distributed (backend nccl == RCCL on ROCm, gloo on CPU).  Runs the
    collective whenever a process group exists -- also a group of one rank; without a group (a plain single-process run) it is
    the identity.  The device-resident exchange is ShardedMsm."""
    part = np.ascontiguousarray(partial, np.uint8).reshape(PARTIAL_BYTES)
    if not _pg_active():
        if world != 1:
            raise RuntimeError("allgather_partials: world > 1 without a process group")
        return part.copy()
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(part.copy())
    if device is not None:
        mine = mine.to(device)
    out = torch.empty(world * PARTIAL_BYTES, dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().numpy()


def sum_partials(partials: np.ndarray, k: int) -> bytes:
    """curve addition of k partials + normalisation -> 96 canonical bytes"""
    p = np.ascontiguousarray(partials, np.uint8)
    out = C.create_string_buffer(96)
    _lib.check(_lib.lib().sonic_g1_sum_partials(p.ctypes.data, k, out))
    return out.raw


def exchange_layout(srs, world: int) -> Tuple[int, int]:
    """(buckets of the shared bucket set, slice length S) for a bucket exchange over `world` ranks"""
    nb, s = C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().sonic_msm_exchange_layout(srs._h, world, C.byref(nb), C.byref(s)))
    return nb.value, s.value


class ShardedMsm:
    """One rank's end of MSMs that are split over the ranks of the default process group (or of a single process when no
    group exists).  Owns a HIP stream (torch), a lane on that stream, and the device tensors the collectives move.

    device: torch cuda device of this rank.  staged=None: collectives on device tensors iff the backend is nccl."""

    def __init__(self, srs, rank: int, world: int, device, staged: Optional[bool] = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.srs, self.rank, self.world, self.device = srs, rank, world, device
        self.pg = _pg_active()
        if world > 1 and not self.pg:
            raise RuntimeError("ShardedMsm: world > 1 needs an initialised process group")
        if staged is None:
            staged = self.pg and dist.get_backend() != "nccl"
        self.staged = staged
        self.stream = torch.cuda.Stream(device=device)
        self._lane = C.c_void_p()
        _lib.check(_lib.lib().sonic_msm_lane_new_on_stream(C.c_void_p(self.stream.cuda_stream), C.byref(self._lane)))
        with torch.cuda.stream(self.stream):          # allocated and cleared on the stream everything else here is ordered on
            self.part = torch.zeros(DEV_PARTIAL_BYTES, dtype=torch.uint8, device=device)
            self.gathered = torch.zeros(world * DEV_PARTIAL_BYTES, dtype=torch.uint8, device=device)
        self.buckets = self.recv = None
        self.n_buckets = self.slice_len = 0
        self._layout_world = 0

    # -- shared tail: all-gather of the 192-byte partials (device tensors, or staged through the host for gloo), ONE copy to the host
    def _gather_and_sum(self) -> bytes:
        torch, dist = self.torch, self.dist
        if self.pg and not self.staged:
            dist.all_gather_into_tensor(self.gathered, self.part)
            host = self.gathered.cpu()                                   # the one device-to-host copy (waits for the stream)
        elif self.pg:
            mine = self.part.cpu()
            host = torch.empty(self.world * DEV_PARTIAL_BYTES, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, mine)
        else:
            host = self.part.cpu()
        _lib.check(_lib.lib().sonic_msm_lane_sync(self._lane))            # the stream has drained: reports non-canonical scalars
        return sum_dev_partials(host.numpy(), self.world)

    def run_terms(self, basis: int, e0: int, d_scalars, n: int) -> bytes:
        """this rank's term slice [e0, e0 + n) through the whole bucket method; partials gathered on the device"""
        L = _lib.lib()
        submitted = False
        with self.torch.cuda.stream(self.stream):
            if n > 0:
                _lib.check(L.sonic_msm_submit_dev_v2(self._lane, self.srs._h, basis, e0, d_scalars, n, C.c_void_p(self.part.data_ptr()), DEV_PARTIAL_BYTES))
                submitted = True
            else:
                self.part.zero_()               # W = 0: the empty sum
            try:
                res = self._gather_and_sum()
            finally:
                # closes the submit whatever the gather reported (a non-canonical scalar raises out of lane_sync): a lane left
                # in flight would hand the NEXT run_terms the result of this one
                if submitted:
                    L.sonic_msm_collect(self._lane, None, None)
        return res

    def _ensure_exchange(self, world: int):
        if self.buckets is None or self.buckets.numel() != world * self.slice_len * PARTIAL_BYTES or self._layout_world != world:
            self.n_buckets, self.slice_len = exchange_layout(self.srs, world)
            self._layout_world = world
            nbytes = world * self.slice_len * PARTIAL_BYTES
            with self.torch.cuda.stream(self.stream):
                self.buckets = self.torch.zeros(nbytes, dtype=self.torch.uint8, device=self.device)
                self.recv = self.torch.zeros(nbytes, dtype=self.torch.uint8, device=self.device)

    def run_buckets(self, basis: int, e0: int, d_scalars, n: int) -> bytes:
        """this rank's term slice accumulated into a full bucket set, bucket ranges exchanged (all-to-all), 1/world of the
        buckets reduced here, partials gathered"""
        L, torch, dist = _lib.lib(), self.torch, self.dist
        self._ensure_exchange(self.world)
        S = self.slice_len
        with torch.cuda.stream(self.stream):
            rc = L.sonic_msm_accumulate_dev(self._lane, self.srs._h, basis, e0, d_scalars, n, C.c_void_p(self.buckets.data_ptr()), self.world * S)
            err = _lib.last_error() if rc else ""
            if self.pg:
                # a rank that cannot take part (bad arguments, a slice too large) must not leave the others waiting in the
                # all-to-all: agree first, on the host (one 4-byte all-reduce; the exchange that follows moves megabytes)
                ok = torch.tensor([0 if rc else 1], dtype=torch.int32, device=self.part.device if not self.staged else "cpu")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    raise _lib.SonicError(rc or 7, err or "run_buckets: another rank could not accumulate its slice")
            elif rc:
                raise _lib.SonicError(rc, err)
            if self.pg and not self.staged:
                dist.all_to_all_single(self.recv, self.buckets)           # equal splits: S x 192 B per pair
            elif self.pg:
                src = self.buckets.cpu()
                dst = torch.empty_like(src)
                dist.all_to_all_single(dst, src)
                self.recv.copy_(dst)
            else:
                self.recv.copy_(self.buckets)
            _lib.check(L.sonic_msm_reduce_slices_dev_v2(self._lane, self.srs._h, C.c_void_p(self.recv.data_ptr()), self.world, S,
                                                        self.rank * S, C.c_void_p(self.part.data_ptr()), DEV_PARTIAL_BYTES))
            return self._gather_and_sum()

    def run_buckets_emulated(self, basis: int, e0: int, d_scalars, n: int, world_emul: int, rank_emul: int = 0) -> None:
        """TIMING ONLY (one GPU standing in for one of `world_emul` ranks): the accumulation of a 1/world share of the terms, a
        device-to-device copy of as many bytes as this rank's all-to-all would send and receive, the element-wise addition of
        world_emul slices and the reduction of 1/world_emul of the buckets.  The slices added are this rank's own, so the
        value is meaningless; the work and the traffic are those of the real exchange minus the xGMI hop."""
        L, torch = _lib.lib(), self.torch
        self._ensure_exchange(world_emul)
        S = self.slice_len
        with torch.cuda.stream(self.stream):
            _lib.check(L.sonic_msm_accumulate_dev(self._lane, self.srs._h, basis, e0, d_scalars, n,
                                                  C.c_void_p(self.buckets.data_ptr()), world_emul * S))
            self.recv.copy_(self.buckets)
            _lib.check(L.sonic_msm_reduce_slices_dev_v2(self._lane, self.srs._h, C.c_void_p(self.recv.data_ptr()), world_emul, S,
                                                        rank_emul * S, C.c_void_p(self.part.data_ptr()), DEV_PARTIAL_BYTES))
            self.part.cpu()
        _lib.check(L.sonic_msm_lane_sync(self._lane))

    def close(self):
        if self._lane:
            _lib.lib().sonic_msm_lane_free(self._lane)
            self._lane = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allgather_shares(share: bytes, world: int, device=None):
    """all-gather of one proof share per rank over the default process group: device tensors over RCCL (backend nccl), host
    tensors over gloo; returns the `world` shares in rank order"""
    import torch
    import torch.distributed as dist
    staged = dist.get_backend() != "nccl"
    mine = torch.frombuffer(bytearray(share), dtype=torch.uint8)
    if not staged:
        mine = mine.to(device)
    gathered = torch.empty(world * len(share), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine)
    host = gathered.cpu().numpy().tobytes()
    return [host[i * len(share):(i + 1) * len(share)] for i in range(world)]


class ShardedProver:
    """ONE proof over the ranks of the default process group (BASELINE configs[3] read as "a single n = 2^20 instance on 8 GPUs").

    Every rank constructs it with the same circuit over its replica of the SRS and calls set_assignment / prove_bytes with the same
    assignment and transcript; prove_bytes returns the same proof bytes on every rank, byte-identical to Prover.prove_bytes on one
    GPU.  The only collective is one all-gather of sonic_proof_share_size(Q) bytes per rank (device tensors over RCCL, host tensors
    over gloo).  Without a process group it is a plain Prover.

    emulate=(rank, world): no collective -- this process runs the share of `rank` of `world` (timing one rank's work on one GPU,
    and the tests that run every rank's share in turn and combine them)."""

    def __init__(self, srs, circuit, rank: int = 0, world: int = 1, device=None, prepare: bool = True, emulate=None):
        from .protocol import Prover
        self.rank, self.world, self.device = rank, world, device
        self.emulate = emulate
        self.pg = _pg_active()
        if world > 1 and not self.pg and emulate is None:
            raise RuntimeError("ShardedProver: world > 1 needs an initialised process group")
        self.prover = Prover(srs, circuit, prepare=prepare)
        self.Q = self.prover.Q
        if emulate is not None:
            self.prover.set_share(*emulate)
        elif world > 1:
            self.prover.set_share(rank, world)
        self.share_bytes = _lib.lib().sonic_proof_share_size(self.Q)

    def set_assignment(self, assignment):
        self.prover.set_assignment(assignment)

    def set_emulated_rank(self, rank: int, world: int):
        self.emulate = (rank, world)
        self.prover.set_share(rank, world)

    def prove_share(self, transcript) -> bytes:
        """this rank's share only (no collective)"""
        return self.prover.prove_share(transcript)

    def prove_bytes(self, transcript) -> bytes:
        from .protocol import proof_from_shares
        if self.emulate is not None:
            raise RuntimeError("ShardedProver: an emulated rank has only a share (prove_share)")
        if not self.pg:
            return self.prover.prove_bytes(transcript)
        # (a group of ONE rank still goes through the share and the all-gather: the collective runs wherever a group exists)
        # a rank whose share fails must still enter the all-gather (the others would hang in it): it reports through the flags
        # of its share when the library got that far, else re-raises after the collective
        err = None
        try:
            share = self.prover.prove_share(transcript)
        except Exception as e:      # noqa: BLE001
            err, share = e, bytes(self.share_bytes)
        shares = allgather_shares(share, self.world, self.device)
        if err is not None:
            raise err
        return proof_from_shares(self.Q, shares, transcript)

    def close(self):
        self.prover.close()
