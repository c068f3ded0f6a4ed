"""The N > 1 path on a GPU box with one GPU: two ranks (gloo) share the device.  Each rank reduces its slice of a range-sharded
MSM with the real kernels (sonic_msm_g1_srs_partial_dev), the 192-byte partials are all-gathered and summed, and the result must
be the oracle's full MSM -- and so must the two schemes of sonic_amd.distributed.ShardedMsm: term ranges with the partials
gathered, and term ranges + the all-to-all of bucket ranges (both staged through host tensors here, because two RCCL ranks cannot
share one device; tests/test_gpu_configs.py runs the same code over RCCL with one rank).  At 2^21 terms per rank (an N = 2^22 MSM,
BASELINE.json configs[3] on two ranks) the reference value is the closed-form trapdoor exponent of a geometric scalar vector.
Then bench.py itself is run as the driver runs it for N = 2 (torch.distributed.run)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _reports(fn):
    """a worker that fails puts its exception on the queue, so that the parent fails at once instead of waiting out its timeout"""
    import functools

    @functools.wraps(fn)
    def run(rank, *args):
        try:
            fn(rank, *args)
        except BaseException as e:      # noqa: BLE001
            args[-1].put((rank, False, repr(e)))
            raise
    return run


@_reports
def _worker_big(rank, world, port, q):
    """N = 2^22 over two ranks: 2^21 terms each, SRS d = 2^21 on both (replicated), scalars 1, x0, x0^2, ..."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import ctypes as C
    import torch
    import torch.distributed as dist
    import sonic_amd
    from oracle import orc
    from sonic_amd import _lib, distributed as sd
    from util import R
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = _lib.lib()
        _lib.check(L.sonic_init(0))
        d, x, alpha, x0 = 1 << 21, 0x123456789abcdef, 0xfedcba987654321, 0x1000000000000007
        N = 1 << 22
        srs = sonic_amd.SRS.new(d, x, alpha)
        lo, hi = sd.split_range(N, world, rank)
        buf = bytearray(32 * (hi - lo))
        v = pow(x0, lo, R)
        for i in range(hi - lo):
            buf[32 * i:32 * i + 32] = v.to_bytes(32, "little")
            v = v * x0 % R
        mine = np.frombuffer(bytes(buf), np.uint8)
        dsc = C.c_void_p()
        _lib.check(L.sonic_dev_alloc(mine.size, C.byref(dsc)))
        _lib.check(L.sonic_dev_upload(dsc, mine.ctypes.data, mine.size))
        xx = x0 * x % R
        want = orc.g1_mul(orc.g1_gen(), pow(x, -d, R) * (pow(xx, N, R) - 1) % R * pow(xx - 1, -1, R) % R)
        sh = sd.ShardedMsm(srs, rank, world, torch.device("cuda", 0))
        ok_terms = sh.run_terms(0, -d + lo, dsc, hi - lo) == want
        ok_buckets = sh.run_buckets(0, -d + lo, dsc, hi - lo) == want
        sh.close()
        L.sonic_dev_free(dsc)
        q.put((rank, ok_terms and ok_buckets))
    finally:
        dist.destroy_process_group()


def test_msm_2p22_two_ranks_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_big, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


@_reports
def _worker(rank, world, port, n_terms, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import ctypes as C
    import torch.distributed as dist
    import sonic_amd
    from oracle import orc
    from sonic_amd import _lib, distributed as sd
    from util import rand_fr_array
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = _lib.lib()
        _lib.check(L.sonic_init(0))
        d, x, alpha = 1 << 12, 0x1234567, 0x7654321
        srs = sonic_amd.SRS.new(d, x, alpha)
        sc = rand_fr_array(np.random.default_rng(4242), n_terms)
        e0 = -d + 5
        lo, hi = sd.split_range(n_terms, world, rank)
        part = np.zeros(192, np.uint8)
        if hi > lo:
            dsc = C.c_void_p()
            mine = np.ascontiguousarray(sc[lo:hi])
            _lib.check(L.sonic_dev_alloc(32 * (hi - lo), C.byref(dsc)))
            _lib.check(L.sonic_dev_upload(dsc, mine.ctypes.data, 32 * (hi - lo)))
            _lib.check(L.sonic_msm_g1_srs_partial_dev(srs._h, 0, e0 + lo, dsc, hi - lo, part.ctypes.data))
            L.sonic_dev_free(dsc)
        got = sd.sum_partials(sd.allgather_partials(part, world), world)
        osrs = orc.SRS(d, x, alpha, threads=4)
        want = orc.msm_srs(osrs, 0, e0, sc, 1, 4)
        # the same through ShardedMsm: partials gathered from device tensors, and the bucket exchange (staged: gloo)
        import torch
        sh = sd.ShardedMsm(srs, rank, world, torch.device("cuda", 0))
        assert sh.staged
        dmine = C.c_void_p()
        mine = np.ascontiguousarray(sc[lo:hi]) if hi > lo else np.zeros((1, 32), np.uint8)
        _lib.check(L.sonic_dev_alloc(mine.size, C.byref(dmine)))
        _lib.check(L.sonic_dev_upload(dmine, mine.ctypes.data, mine.size))
        ok2 = sh.run_terms(0, e0 + lo, dmine, hi - lo) == want
        ok3 = sh.run_buckets(0, e0 + lo, dmine, hi - lo) == want
        sh.close()
        q.put((rank, got == want and ok2 and ok3))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_terms,world", [(5000, 2), (3, 2), (5000, 4), (3, 4)])       # 3 terms over 4 ranks: two ranks have nothing
def test_range_sharded_msm_ranks_share_one_gpu(n_terms, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_terms, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_bench_two_ranks_one_gpu():
    """the driver's N = 2 launch line, small sizes, gloo: one JSON line from rank 0 with whole-job values"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
           "--warmup", "1", "--log2n", "10", "--msm-log2", "12", "--no-cpu", "--strong-log2n", "12"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0 and j["msm"]["value"] > 0
    assert j["roofline"]["bound"] == "hbm" and j["cpu_baseline"] is None
    assert j["msm_strong"]["scaling"] == "strong" and j["msm_strong"]["n_gpus"] == 2 and j["msm_strong"]["same_result_as_term_range_sharding"]
    assert "all-to-all" in j["msm_strong"]["method"]
    # ONE proof shared by the two ranks, byte-equal to the proof rank 0 makes alone
    ps = j["prove_strong"]
    assert ps["scaling"] == "strong" and ps["n_gpus"] == 2 and ps["n"] == 1 << 12 and ps["same_bytes_as_one_gpu_alone"] is True and ps["ms_per_proof"] > 0
    assert j["north_star"] is None
    # rank 0 also drove "both GPUs" (here: the one GPU twice) from its one process through the C ABI while rank 1 waited on the store
    ip = j["in_process"]
    assert ip["devices"] == [0, 0] and ip["proofs"]["same_bytes_as_one_handle_alone"] is True and ip["prove_strong"]["same_bytes_as_one_gpu_alone"] is True
    assert ip["msm_strong"]["bucket_ranges"]["same_result_as_one_gpu"] is True and j["status"] == "ok"


def test_bench_starts_its_own_ranks_and_never_mislabels():
    """`python bench.py --gpus N` WITHOUT a launcher: it starts the N ranks itself before touching the GPU and
    relays rank 0's line (gloo here: two ranks on the one GPU); asked for more RCCL ranks than the node has GPUs it refuses with a
    non-zero status instead of measuring one rank under the label N; the in-process form (one process, the C ABI, a device list)"""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    small = ["--steps", "2", "--warmup", "1", "--log2n", "10", "--msm-log2", "12", "--no-cpu", "--strong-log2n", "11", "--msm-strong-log2", "13", "--no-sensitivities"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"] + small, cwd=ROOT, capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["process_group"].startswith("gloo, 2 rank") and j["status"] == "ok" and j["prove_strong"]["n_gpus"] == 2
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")] and "not measuring" in out.stderr
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--in-process", "--devices", "0,0,0"] + small, cwd=ROOT,
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    ip = j["in_process"]
    assert j["n_gpus"] == 3 and j["value"] > 0 and j["status"] == "ok" and "ONE process" in j["config"]["process_group"]
    assert ip["proofs"]["same_bytes_as_one_handle_alone"] is True and ip["prove_strong"]["same_bytes_as_one_gpu_alone"] is True
    assert ip["msm_strong"]["bucket_ranges"]["same_result_as_one_gpu"] is True and ip["msm_strong"]["term_ranges"]["same_result_as_one_gpu"] is True


@_reports
def _worker_rccl(rank, world, port, q):
    """one rank per GPU over RCCL: the device-tensor collectives of ShardedMsm (all_gather_into_tensor, all_to_all_single ordered on
    the lane's stream) and the share all-gather of ShardedProver, against closed forms / the single-GPU proof"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import ctypes as C
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    try:
        import sonic_amd
        from oracle import orc
        from sonic_amd import _lib, distributed as sd
        from util import R, big_circuit, rand_fr_array
        L = _lib.lib()
        _lib.check(L.sonic_init(rank))
        d, x, alpha, x0 = 1 << 17, 0x123456789abcdef, 0xfedcba987654321, 0x1000000000000007
        N = 1 << 18
        srs = sonic_amd.SRS.new(d, x, alpha)
        lo, hi = sd.split_range(N, world, rank)
        buf = bytearray(32 * (hi - lo))
        v = pow(x0, lo, R)
        for i in range(hi - lo):
            buf[32 * i:32 * i + 32] = v.to_bytes(32, "little")
            v = v * x0 % R
        mine = np.frombuffer(bytes(buf), np.uint8)
        dsc = C.c_void_p()
        _lib.check(L.sonic_dev_alloc(mine.size, C.byref(dsc)))
        _lib.check(L.sonic_dev_upload(dsc, mine.ctypes.data, mine.size))
        xx = x0 * x % R
        want = orc.g1_mul(orc.g1_gen(), pow(x, -d, R) * (pow(xx, N, R) - 1) % R * pow(xx - 1, -1, R) % R)
        sh = sd.ShardedMsm(srs, rank, world, device)
        assert not sh.staged
        ok = sh.run_terms(0, -d + lo, dsc, hi - lo) == want
        ok = ok and sh.run_buckets(0, -d + lo, dsc, hi - lo) == want
        ok = ok and sh.run_terms(0, -d + lo, dsc, hi - lo) == want          # and again after an exchange (buffers reused)
        sh.close()
        L.sonic_dev_free(dsc)
        # one proof over the ranks
        n, Q = 1 << 12, 2
        circ = big_circuit(21, n, Q)
        circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
        asg = sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"])
        tr = rand_fr_array(np.random.default_rng(8), 8 + 2 * Q)
        tr[:, 0] |= 1
        sp = sd.ShardedProver(srs, circuit, rank, world, device)
        sp.set_assignment(asg)
        got = sp.prove_bytes(tr)
        sp.close()
        one = sonic_amd.Prover(srs, circuit, prepare=False)
        one.set_assignment(asg)
        ok = ok and one.prove_bytes(tr) == got
        one.close()
        q.put((rank, ok, got[:16].hex()))
    finally:
        dist.destroy_process_group()


def _gpu_count():
    import torch
    return torch.cuda.device_count()          # counting does not initialise the GPU in this process


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL cannot put two ranks on one device (the pool's test boxes have one; "
                                             "the same paths run there over gloo and over a one-rank RCCL group)")
def test_rccl_ranks_on_their_own_gpus():
    world = min(_gpu_count(), 8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rccl, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(120)
    assert all(r[1] for r in res), res
    assert len({r[2] for r in res}) == 1
