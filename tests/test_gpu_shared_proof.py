"""ONE proof over several ranks on the GPU (sonic_prover_set_share; SURVEY 8e "MSM-level parallelism", BASELINE configs[3] as a
single instance on 8 GPUs).  Every rank's share is computed by the real kernels -- all ranks in turn on one handle, then as 2 and 4
gloo processes sharing the one GPU of the box -- and the combined proof must equal, byte for byte, the proof of one GPU and the
CPU oracle's (src/Sonic/Protocol.hs:47-109, src/Sonic/Signature.hs:38-72)."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from util import R, circuit_arrays, fr_bytes, big_circuit, rand_fr_array

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(sonic, orc, ref, n, Q, seed, d=None):
    pyr = random.Random(seed)
    d = d or max(8 * n, 12)          # n = 1 needs d >= 4n + 8 (t(X,y) reaches X^{-4n-8})
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    srs = sonic.SRS.new(d, x, alpha)
    osrs = orc.SRS(d, x, alpha, threads=os.cpu_count() or 1)
    circ, asg, enc = circuit_arrays(ref, pyr, n, Q)
    tr = [pyr.randrange(1, R) for _ in range(8 + 2 * Q)]
    want = orc.prove(osrs, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], fr_bytes(tr))
    circuit = sonic.ArithCircuit(sonic.GateWeights(circ[0], circ[1], circ[2]), circ[3])
    return srs, circuit, sonic.Assignment(*asg), tr, want


@pytest.mark.parametrize("n,Q", [(1, 1), (3, 2), (16, 2), (100, 3), (257, 1), (1000, 4)])
@pytest.mark.parametrize("prepare", [False, True])
def test_all_shares_of_a_proof_combine_to_the_oracle_proof(sonic, orc, ref, n, Q, prepare):
    srs, circuit, asg, tr, want = _setup(sonic, orc, ref, n, Q, 100 + n + Q)
    p = sonic.Prover(srs, circuit, prepare=prepare)
    p.set_assignment(asg)
    assert p.prove_bytes(tr) == want
    for world in (2, 3, 8, 16):               # (16 ranks: more than a small proof has pieces for -- some ranks report empty shares)
        shares = []
        for r in range(world):
            p.set_share(r, world)
            shares.append(p.prove_share(tr))
        assert sonic.proof_from_shares(Q, shares, tr) == want, (n, Q, world)
    # streamed form (submit / collect_share), and back to the whole proof on the same handle
    p.set_share(1, 2)
    p.submit(tr)
    s1 = p.collect_share()
    p.set_share(0, 2)
    p.submit(tr)
    assert sonic.proof_from_shares(Q, [p.collect_share(), s1], tr) == want
    with pytest.raises(sonic.SonicError):
        p.prove_bytes(tr)                    # a share handle does not hand out whole proofs
    p.set_share(0, 1)
    assert p.prove_bytes(tr) == want
    p.close()


def test_shares_with_cut_msms_at_2p14(sonic, orc, ref, monkeypatch):
    """n = 2^14 on an SRS with window tables (d = 2^17): the plan cuts inside MSMs, the pieces run over the tables.  With the fixed
    parts of the cost model set to zero the line is cut into equal term counts, so every world size cuts some MSM."""
    for k in ("JOB", "R1", "SY", "SU", "T"):
        monkeypatch.setenv("SONIC_SHARE_COST_" + k, "0.0001")
    n, Q = 1 << 14, 2
    circ = big_circuit(7, n, Q)
    x, alpha = 0x1234567891, 0x9876543211
    d = 8 * n
    srs = sonic.SRS.new(d, x, alpha)
    circuit = sonic.ArithCircuit(sonic.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
    tr = rand_fr_array(np.random.default_rng(3), 8 + 2 * Q)
    tr[:, 0] |= 1
    p = sonic.Prover(srs, circuit, prepare=True)
    p.set_assignment(sonic.Assignment(circ["aL"], circ["aR"], circ["aO"]))
    whole = p.prove_bytes(tr)
    osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1))
    assert orc.prove(osrs, n, Q, circ["wL"], circ["wR"], circ["wO"], circ["cs"], circ["aL"], circ["aR"], circ["aO"], tr, True) == whole
    nb, w = srs_plan(sonic, srs, 3 * n)
    for world in (2, 4, 8):
        plan = sonic.share_plan(n, Q, True, world, nb=nb, w=w)
        assert any(0 < lo or hi < (1 << 20) for pieces, _ in plan for lo, hi in pieces if hi > lo)      # at least one MSM is cut
        shares = []
        for r in range(world):
            p.set_share(r, world)
            shares.append(p.prove_share(tr))
        assert sonic.proof_from_shares(Q, shares, tr) == whole, world
    p.close()


def srs_plan(sonic, srs, n_terms):
    import ctypes as C
    from sonic_amd import _lib
    c, w, sets = C.c_int(), C.c_int(), C.c_int()
    _lib.check(_lib.lib().sonic_msm_plan(srs._h, n_terms, C.byref(c), C.byref(w), C.byref(sets)))
    return 1 << (c.value - 1), w.value


def test_share_error_contract(sonic, orc, ref):
    """an unsatisfied circuit (t(X,y) has a constant term: `index` panics in the reference, Protocol.hs:73 via
    CommitmentScheme.hs:70-73) surfaces from the combine with the status sonic_prove returns, whichever rank saw it"""
    n, Q = 16, 2
    srs, circuit, asg, tr, _ = _setup(sonic, orc, ref, n, Q, 77)
    bad = sonic.Assignment(list(asg.aL), list(asg.aR), [(v + 1) % R for v in asg.aO])
    p = sonic.Prover(srs, circuit, prepare=False)
    p.set_assignment(bad)
    with pytest.raises(sonic.SonicError) as e0:
        p.prove_bytes(tr)
    shares = []
    for r in range(4):
        p.set_share(r, 4)
        shares.append(p.prove_share(tr))
    with pytest.raises(sonic.SonicError) as e1:
        sonic.proof_from_shares(Q, shares, tr)
    assert e1.value.code == e0.value.code == 2
    # the Fiat-Shamir chain serialises the MSMs: not available on a share handle
    with pytest.raises(sonic.SonicError):
        p.prove_fs(bytes(32), bytes(32))
    p.close()


def _worker(rank, world, port, n, Q, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import torch
        import torch.distributed as dist
        import sonic_amd
        from oracle import orc
        from sonic_amd import _lib, distributed as sd
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            _lib.check(_lib.lib().sonic_init(0))
            circ = big_circuit(21, n, Q)
            x, alpha, d = 0x1234567891, 0x9876543211, 8 * n
            srs = sonic_amd.SRS.new(d, x, alpha)
            circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
            asg = sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"])
            tr = rand_fr_array(np.random.default_rng(8), 8 + 2 * Q)
            tr[:, 0] |= 1
            sp = sd.ShardedProver(srs, circuit, rank, world, torch.device("cuda", 0))
            sp.set_assignment(asg)
            got = sp.prove_bytes(tr)
            got2 = sp.prove_bytes(tr)
            sp.close()
            ok = got == got2
            if rank == 0:
                one = sonic_amd.Prover(srs, circuit, prepare=False)
                one.set_assignment(asg)
                ok = ok and one.prove_bytes(tr) == got
                one.close()
                osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1))
                ok = ok and orc.prove(osrs, n, Q, circ["wL"], circ["wR"], circ["wO"], circ["cs"], circ["aL"], circ["aR"], circ["aO"], tr, True) == got
            q.put((rank, ok, got[:16].hex()))
        finally:
            dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        q.put((rank, False, repr(e)))


@pytest.mark.parametrize("world,n", [(2, 1 << 12), (4, 1 << 12), (4, 50)])
def test_one_proof_over_gloo_ranks_sharing_the_gpu(world, n):
    """ShardedProver end to end: `world` processes (gloo; they share the box's one GPU), one all-gather of the shares, the same
    proof bytes on every rank, equal to the single-GPU proof and to the oracle's"""
    Q = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, Q, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({h for _, _, h in res}) == 1
