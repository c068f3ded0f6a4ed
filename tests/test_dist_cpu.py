"""The N > 1 path on CPU: world_size-2 gloo.  Each rank takes its slice of a range-sharded MSM, the
192-byte partials are all-gathered, and every rank adds them with the library's own (host-side)
sonic_g1_sum_partials.  Without a GPU the per-rank bucket sums come from the oracle -- this test covers
the sharding plan, the exchange and the curve-addition combine, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def _affine_to_partial(b96: bytes) -> np.ndarray:
    """canonical affine bytes -> the library's XYZZ partial (Montgomery limbs): (x, y, 1, 1) or all-zero"""
    if b96 == bytes(96):
        return np.zeros(192, np.uint8)
    Rm = 1 << 384
    x = int.from_bytes(b96[:48], "little") * Rm % Q_MOD
    y = int.from_bytes(b96[48:], "little") * Rm % Q_MOD
    one = Rm % Q_MOD
    return np.frombuffer(b"".join(v.to_bytes(48, "little") for v in (x, y, one, one)), np.uint8).copy()


def _worker(rank, world, port, n_terms, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import orc
    from sonic_amd import distributed as sd
    from util import rand_fr_array
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = 300
        srs = orc.SRS(d, 11, 13, threads=2)
        sc = rand_fr_array(np.random.default_rng(42), n_terms)
        e0 = -200
        lo, hi = sd.split_range(n_terms, world, rank)
        mine = orc.msm_srs(srs, 0, e0 + lo, sc[lo:hi], 1, 2) if hi > lo else bytes(96)
        parts = sd.allgather_partials(_affine_to_partial(mine), world)
        got = sd.sum_partials(parts, world)
        want = orc.msm_srs(srs, 0, e0, sc, 1, 2)
        q.put((rank, got == want, sd.msm_shard(rank, world, d, 100)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_terms", [401, 3])
def test_range_sharded_msm_world2(n_terms):
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sonic_amd", "csrc"), "-s", "-j8"])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_terms, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    shards = dict((r, s_) for r, _, s_ in res)
    assert shards[0] == (0, -300) and shards[1] == (0, -200)


def test_shard_plan():
    sys.path.insert(0, ROOT)
    from sonic_amd import distributed as sd
    d, per = 1 << 21, 1 << 20
    plan = [sd.msm_shard(r, 8, d, per) for r in range(8)]
    assert plan[:4] == [(0, -d + k * per) for k in range(4)] and plan[4:] == [(1, -d + k * per) for k in range(4)]
    with pytest.raises(ValueError):
        sd.msm_shard(8, 16, d, per)
    assert [sd.split_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sd.split_range(2, 4, 3) == (2, 2)
