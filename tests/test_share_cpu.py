"""ONE proof over several ranks, the parts that need no GPU: the plan (sonic_prove_share_plan: every MSM's terms are covered exactly
once, the modelled cost is balanced), the share format and the combine (sonic_proof_from_shares), and the exchange over a
world_size-2 gloo group.  Without a GPU the MSM pieces cannot be computed, so the shares are synthesised from the oracle's proof:
the rank whose piece of an MSM starts at term 0 carries the whole point, every other piece the point at infinity -- what is tested
is the plan, the blob, the all-gather and the curve-addition combine, not the kernels (tests/test_gpu_shared_proof.py)."""
import ctypes as C
import os
import random
import socket
import struct
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
Q_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
ONE = 1 << 20


def _partial(b96: bytes) -> bytes:
    """canonical affine bytes -> the library's 192-byte XYZZ partial (Montgomery limbs): (x, y, 1, 1), or zeros for infinity"""
    if b96 == bytes(96):
        return bytes(192)
    Rm = 1 << 384
    x = int.from_bytes(b96[:48], "little") * Rm % Q_MOD
    y = int.from_bytes(b96[48:], "little") * Rm % Q_MOD
    one = Rm % Q_MOD
    return b"".join(v.to_bytes(48, "little") for v in (x, y, one, one))


def proof_parts(proof: bytes, Q: int):
    """proof bytes -> (points by slot, evaluations a, b, s, s_j, s'_j): the inverse of the layout in include/sonic_hip.h"""
    K, F = 7 + 4 * Q, 3 + 2 * Q
    pts, frs = [None] * K, [None] * F
    pos = 0

    def g(slot):
        nonlocal pos
        pts[slot] = proof[pos:pos + 96]; pos += 96

    def f(i):
        nonlocal pos
        frs[i] = proof[pos:pos + 32]; pos += 32
    g(0); g(1); f(0); g(2); f(1); g(3); g(4); f(2)
    for j in range(Q):
        g(5 + 2 * j); f(3 + j); g(6 + 2 * j)
    for j in range(Q):
        f(3 + Q + j); g(5 + 2 * Q + 2 * j); g(6 + 2 * Q + 2 * j)
    g(5 + 4 * Q); g(6 + 4 * Q)
    return pts, frs


# which slot's first piece reports which evaluation (prove.hip): a <- W_a, b <- W_b, s <- W_t, s_j <- W_j, s'_j <- Q_j
def fr_owner_slot(i, Q):
    return [2, 3, 4][i] if i < 3 else (6 + 2 * (i - 3) if i < 3 + Q else 6 + 2 * Q + 2 * (i - 3 - Q))


def synth_share(proof: bytes, Q: int, rank: int, world: int, pieces, flags: int = 0, plan_tag: int = 0x1234, version: int = 2) -> bytes:
    K, F = 7 + 4 * Q, 3 + 2 * Q
    pts, frs = proof_parts(proof, Q)
    out = struct.pack("<IIiiqii", 0x48534E53, version, rank, world, Q, flags, plan_tag)
    out += b"".join(struct.pack("<II", lo, hi) for lo, hi in pieces)
    out += b"".join(_partial(pts[i]) if (pieces[i][1] > pieces[i][0] and pieces[i][0] == 0) else bytes(192) for i in range(K))
    valid = [int(pieces[fr_owner_slot(i, Q)][1] > 0 and pieces[fr_owner_slot(i, Q)][0] == 0) for i in range(F)]
    out += b"".join(frs[i] if valid[i] else bytes(32) for i in range(F))
    out += b"".join(struct.pack("<i", v) for v in valid)
    return out


@pytest.mark.parametrize("n,Q,prepared", [(1 << 20, 2, True), (1 << 18, 2, False), (1 << 18, 1, True), (1 << 16, 5, True), (16, 2, False), (1, 1, False),
                                          (257, 3, True), (1 << 14, 40, True)])
def test_plan_covers_every_msm_once_and_is_balanced(n, Q, prepared):
    import sonic_amd
    from sonic_amd import _lib
    assert _lib.lib().sonic_proof_share_size(Q) == 32 + (7 + 4 * Q) * 200 + (3 + 2 * Q) * 36
    K = 7 + 4 * Q
    for world in (1, 2, 3, 4, 8, 16):
        plan = sonic_amd.share_plan(n, Q, prepared, world)
        for slot in range(K):
            pcs = sorted((lo, hi) for pieces, _ in plan for lo, hi in [pieces[slot]] if hi > lo)
            at = 0
            for lo, hi in pcs:
                assert lo == at, (n, Q, world, slot, pcs)
                at = hi
            assert at == ONE, (n, Q, world, slot, pcs)
        costs = [c for _, c in plan]
        if n >= 1 << 16 and world <= 8:
            terms = (45 if prepared else 49) * n if Q == 2 else None
            # the slowest rank is within 25 % of the mean (fixed parts included) ...
            assert max(costs) <= 1.25 * sum(costs) / world, (n, Q, world, costs)
            # ... and, for the bench shape, within 30 % of the MSM terms alone divided evenly (the price of the repeated polynomials)
            if terms and world == 8 and n >= 1 << 20:       # (at smaller n the per-piece bucket reduction weighs more)
                assert max(costs) <= 1.30 * terms / world, (costs, terms / world)


def test_plan_is_deterministic_and_contiguous():
    import sonic_amd
    a = sonic_amd.share_plan(1 << 20, 2, True, 8)
    assert a == sonic_amd.share_plan(1 << 20, 2, True, 8)
    # every rank gets work at this size, and no rank more than two cut MSMs (one at each end of its stretch of the line)
    for pieces, cost in a:
        assert cost > 0
        assert sum(1 for lo, hi in pieces if hi > lo and (lo, hi) != (0, ONE)) <= 2


def _oracle_proof(n, Q, seed):
    from oracle import orc, sonic_ref as ref
    from util import R, circuit_arrays, fr_bytes
    pyr = random.Random(seed)
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    osrs = orc.SRS(8 * n, x, alpha, threads=2)
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
    want = orc.prove(osrs, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(tr))
    return want, tr


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_combine_synthetic_shares(world):
    import sonic_amd
    from sonic_amd import _lib
    n, Q = 8, 2
    want, tr = _oracle_proof(n, Q, 11)
    plan = sonic_amd.share_plan(n, Q, False, world, nb=64, w=37)
    shares = [synth_share(want, Q, r, world, plan[r][0]) for r in range(world)]
    assert sonic_amd.proof_from_shares(Q, shares, tr) == want
    assert sonic_amd.proof_from_shares(Q, shares[::-1], tr) == want                      # any rank order
    if world > 1:
        with pytest.raises(_lib.SonicError) as e:                                         # a share missing / repeated
            sonic_amd.proof_from_shares(Q, shares[:-1] + [shares[0]], tr)
        assert e.value.code == 7
        # a piece that does not start where its neighbour ends
        # (the plan may cut only between MSMs at this size: then shift the boundary of a whole piece instead)
        r_cut = next((r for r in range(world) if any(0 < lo < hi for lo, hi in plan[r][0])), None)
        bad = [list(p) for p, _ in plan]
        if r_cut is None:
            r_cut = 0
            i_cut = next(i for i, (lo, hi) in enumerate(bad[0]) if hi > lo)
            bad[r_cut][i_cut] = (bad[r_cut][i_cut][0], bad[r_cut][i_cut][1] - 1)
        else:
            i_cut = next(i for i, (lo, hi) in enumerate(bad[r_cut]) if 0 < lo < hi)
            bad[r_cut][i_cut] = (bad[r_cut][i_cut][0] + 1, bad[r_cut][i_cut][1])
        with pytest.raises(_lib.SonicError) as e:
            sonic_amd.proof_from_shares(Q, [synth_share(want, Q, r, world, bad[r]) for r in range(world)], tr)
        assert e.value.code == 7 and "cover" in e.value.message
    # a rank's error flag (here: the SRS-index flag an unsatisfied circuit raises) is the status of the whole proof
    flagged = [synth_share(want, Q, r, world, plan[r][0], flags=2 if r == world - 1 else 0) for r in range(world)]
    with pytest.raises(_lib.SonicError) as e:
        sonic_amd.proof_from_shares(Q, flagged, tr)
    assert e.value.code == 2
    # garbage is refused, not read
    with pytest.raises(_lib.SonicError):
        sonic_amd.proof_from_shares(Q, [bytes(len(shares[0]))] * world, tr)
    # ranks that planned with different parameters (another MSM plan on one GPU, another SONIC_SHARE_COST_* environment) carry
    # different plan tags: the combine names the cause instead of "do not cover" 
    if world > 1:
        mixed = [synth_share(want, Q, r, world, plan[r][0], plan_tag=0x1234 + (r == world - 1)) for r in range(world)]
        with pytest.raises(_lib.SonicError) as e:
            sonic_amd.proof_from_shares(Q, mixed, tr)
        assert e.value.code == 7 and "planned" in e.value.message and "different parameters" in e.value.message
    # a share in the round-4 format (version 1) is refused
    with pytest.raises(_lib.SonicError):
        sonic_amd.proof_from_shares(Q, [synth_share(want, Q, r, world, plan[r][0], version=1) for r in range(world)], tr)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import torch.distributed as dist
        import sonic_amd
        from sonic_amd import distributed as sd
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            n, Q = 8, 2
            want, tr = _oracle_proof(n, Q, 12)
            plan = sonic_amd.share_plan(n, Q, False, world, nb=64, w=37)
            mine = synth_share(want, Q, rank, world, plan[rank][0])
            shares = sd.allgather_shares(mine, world)
            q.put((rank, sonic_amd.proof_from_shares(Q, shares, tr) == want, None))
        finally:
            dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        q.put((rank, False, repr(e)))


def test_shared_proof_exchange_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
