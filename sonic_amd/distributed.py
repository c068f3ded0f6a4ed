"""Multi-GPU plumbing for the one step of the path that shards inside a single MSM.

sum_i s_i * P_i is a sum of independent terms: rank r takes a contiguous slice of the term range on
its own (replicated) SRS, runs the full bucket method on the slice, and contributes one un-normalised
XYZZ partial (192 bytes).  RCCL has no elliptic-curve reduction op, so the exchange is an all-gather of
the partials followed by k-1 curve additions + one normalisation on every rank
(sonic_g1_sum_partials).  prove() itself shards by proof and needs no collective.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib

PARTIAL_BYTES = 192


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


def allgather_partials(partial: np.ndarray, world: int, device=None) -> np.ndarray:
    """all-gather of the 192-byte partials over torch.distributed (backend nccl == RCCL on ROCm, gloo on CPU)"""
    part = np.ascontiguousarray(partial, np.uint8).reshape(PARTIAL_BYTES)
    if world == 1:
        return part.copy()
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(part.copy())
    if device is not None:
        mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return np.concatenate([o.cpu().numpy() for o in out])


def sum_partials(partials: np.ndarray, k: int) -> bytes:
    """curve addition of k partials + normalisation -> 96 canonical bytes"""
    p = np.ascontiguousarray(partials, np.uint8)
    out = C.create_string_buffer(96)
    _lib.check(_lib.lib().sonic_g1_sum_partials(p.ctypes.data, k, out))
    return out.raw
