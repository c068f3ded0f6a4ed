"""BASELINE.json configs[3] and configs[4] as they are worded, on the one GPU of a test box, plus the threading contract of SURVEY 8(b)
and the RCCL branch:

* configs[3] "n=2^20, d=2^22: G1 MSM sharded across 8 GPUs": an N = 2^22 MSM over an SRS slice computed (i) unsharded, (ii) as 8
  term-range shards whose 192-byte partials are added, (iii) as 8 term-range accumulations whose BUCKET ranges are exchanged
  (the all-to-all done in device memory) and reduced per shard -- all three equal, and equal to the closed-form trapdoor
  value for a geometric scalar vector (the fold being split: src/Sonic/CommitmentScheme.hs:25-29);
* configs[4] "batch of 64 independent proofs at n=2^16 streamed": 64 proofs through a two-handle pipeline, every one
  byte-equal to the sequential handle's, two of them byte-equal to the C oracle's;
* three host threads, each with its own pipeline over ONE SRS handle, and four threads each on its own MSM lane: identical bytes;
* a process group of ONE rank over RCCL (backend nccl): the collectives of sonic_amd/distributed.py and of bench.py execute on
  device tensors."""
import ctypes as C
import json
import os
import random
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from util import NCPU, R, big_circuit, fr_bytes, rand_fr_array

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def geom(a, lo, hi):
    return (pow(a, hi + 1, R) - pow(a, lo, R)) * pow(a - 1, -1, R) % R


def geometric_scalars(x0, n):
    """1, x0, x0^2, ... as canonical bytes [n, 32] (python integers: ~1 s per 2^20)"""
    out = bytearray(32 * n)
    v = 1
    for i in range(n):
        out[32 * i:32 * i + 32] = v.to_bytes(32, "little")
        v = v * x0 % R
    return np.frombuffer(bytes(out), np.uint8).reshape(n, 32)


@pytest.fixture(scope="module")
def big_srs(sonic):
    """d = 2^21: the SRS of the n = 2^18 benchmark; basis 0 holds 2^22 + 1 points"""
    pyr = random.Random(31337)
    d = 1 << 21
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    return d, x, alpha, sonic.SRS.new(d, x, alpha)


def _upload(L, _lib, arr):
    p = C.c_void_p()
    _lib.check(L.sonic_dev_alloc(arr.size, C.byref(p)))
    _lib.check(L.sonic_dev_upload(p, arr.ctypes.data, arr.size))
    return p


@pytest.mark.parametrize("kind,world", [("geometric", 8), ("random", 8), ("random", 3)])
def test_msm_2p22_sharded_on_one_gpu(sonic, orc, big_srs, kind, world):
    import torch
    from sonic_amd import _lib, distributed as sd
    L = _lib.lib()
    d, x, alpha, srs = big_srs
    N = 1 << 22
    if kind == "geometric":
        x0 = 0x1234567 * 0x89abcdef + 5
        sc = geometric_scalars(x0, N)
    else:
        sc = rand_fr_array(np.random.default_rng(2222), N)
    dsc = _upload(L, _lib, np.ascontiguousarray(sc))
    # (i) unsharded
    whole = C.create_string_buffer(96)
    _lib.check(L.sonic_msm_g1_srs_dev(srs._h, 0, -d, dsc, N, whole))
    if kind == "geometric":
        # sum_i x0^i g^{x^{-d+i}} = g^{x^-d sum_i (x0 x)^i}
        want = orc.g1_mul(orc.g1_gen(), pow(x, -d, R) * geom(x0 * x % R, 0, N - 1) % R)
        assert whole.raw == want
    # (ii) 8 term-range shards, partials added
    parts = np.zeros((world, 192), np.uint8)
    for r in range(world):
        lo, hi = sd.split_range(N, world, r)
        assert world != 8 or hi - lo == 1 << 19
        _lib.check(L.sonic_msm_g1_srs_partial_dev(srs._h, 0, -d + lo, C.c_void_p(dsc.value + 32 * lo), hi - lo, parts[r].ctypes.data))
    assert sd.sum_partials(parts, world) == whole.raw
    # (iii) 8 term-range accumulations, bucket ranges exchanged (in device memory), 1/8 of the buckets reduced per shard
    NB, S = sd.exchange_layout(srs, world)
    assert NB == 1 << 19 and S % 16384 == 0 and world * S >= NB and (world - 1) * S < NB and (world != 8 or S == NB // 8)
    dev = torch.device("cuda", 0)
    sh = sd.ShardedMsm(srs, 0, 1, dev)
    full = [torch.zeros(world * S * 192, dtype=torch.uint8, device=dev) for _ in range(world)]
    torch.cuda.synchronize()
    with torch.cuda.stream(sh.stream):
        for r in range(world):
            lo, hi = sd.split_range(N, world, r)
            _lib.check(L.sonic_msm_accumulate_dev(sh._lane, srs._h, 0, -d + lo, C.c_void_p(dsc.value + 32 * lo), hi - lo,
                                                  C.c_void_p(full[r].data_ptr()), world * S))
        parts2 = torch.zeros(world * sd.DEV_PARTIAL_BYTES, dtype=torch.uint8, device=dev)
        for r in range(world):
            # what rank r's all-to-all would deliver: slice r of every rank's bucket set, laid out [world][S]
            recv = torch.cat([full[s][r * S * 192:(r + 1) * S * 192] for s in range(world)])
            _lib.check(L.sonic_msm_reduce_slices_dev_v2(sh._lane, srs._h, C.c_void_p(recv.data_ptr()), world, S, r * S,
                                                        C.c_void_p(parts2.data_ptr() + sd.DEV_PARTIAL_BYTES * r), sd.DEV_PARTIAL_BYTES))
            torch.cuda.current_stream().synchronize()
        host = parts2.cpu().numpy()
    _lib.check(L.sonic_msm_lane_sync(sh._lane))
    assert sd.sum_dev_partials(host, world) == whole.raw
    # the single-process forms of the two schemes (no process group: world 1)
    assert sh.run_terms(0, -d, dsc, N) == whole.raw
    assert sh.run_buckets(0, -d, dsc, N) == whole.raw
    sh.close()
    L.sonic_dev_free(dsc)


def test_bucket_exchange_uneven_world_and_errors(sonic, orc):
    """world = 3 (slices padded to the 16384-bucket quantum, the last one partly beyond the bucket set), a small SRS, the C oracle"""
    import torch
    from sonic_amd import _lib, distributed as sd
    L = _lib.lib()
    d, x, alpha = 1 << 16, 0x1234567, 0x7654321
    srs = sonic.SRS.new(d, x, alpha)
    N, world = 100000, 3
    sc = rand_fr_array(np.random.default_rng(5), N)
    dsc = _upload(L, _lib, sc)
    want = orc.msm_srs(orc.SRS(d, x, alpha, threads=NCPU), 1, -d + 7, sc, 1, NCPU)
    NB, S = sd.exchange_layout(srs, world)
    assert S % 16384 == 0 and world * S >= NB          # here the last slice is all padding
    dev = torch.device("cuda", 0)
    sh = sd.ShardedMsm(srs, 0, 1, dev)
    full = [torch.zeros(world * S * 192, dtype=torch.uint8, device=dev) for _ in range(world)]
    parts = torch.zeros(world * sd.DEV_PARTIAL_BYTES, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(sh.stream):
        for r in range(world):
            lo, hi = sd.split_range(N, world, r)
            _lib.check(L.sonic_msm_accumulate_dev(sh._lane, srs._h, 1, -d + 7 + lo, C.c_void_p(dsc.value + 32 * lo), hi - lo,
                                                  C.c_void_p(full[r].data_ptr()), world * S))
        for r in range(world):
            recv = torch.cat([full[s][r * S * 192:(r + 1) * S * 192] for s in range(world)])
            _lib.check(L.sonic_msm_reduce_slices_dev_v2(sh._lane, srs._h, C.c_void_p(recv.data_ptr()), world, S, r * S,
                                                        C.c_void_p(parts.data_ptr() + sd.DEV_PARTIAL_BYTES * r), sd.DEV_PARTIAL_BYTES))
            torch.cuda.current_stream().synchronize()
        host = parts.cpu().numpy()
    assert sd.sum_dev_partials(host, world) == want
    # contract: capacity, quantum, SRS range, non-canonical scalars
    assert L.sonic_msm_accumulate_dev(sh._lane, srs._h, 1, -d, dsc, N, C.c_void_p(full[0].data_ptr()), NB - 1) == 7
    assert L.sonic_msm_reduce_slices_dev_v2(sh._lane, srs._h, C.c_void_p(full[0].data_ptr()), 1, S - 1, 0, C.c_void_p(parts.data_ptr()), sd.DEV_PARTIAL_BYTES) == 7
    # ABI: a buffer of the old 192 bytes is refused, and so are the retired symbols, which a caller built against an older
    # header would still bind -- a status, never 12 KB written over a 192-byte buffer
    assert L.sonic_msm_reduce_slices_dev_v2(sh._lane, srs._h, C.c_void_p(full[0].data_ptr()), 1, S, 0, C.c_void_p(parts.data_ptr()), 192) == 7
    assert "12304" in _lib.last_error() or "SONIC_G1_DEV_PARTIAL_BYTES" in _lib.last_error()
    assert L.sonic_msm_submit_dev_v2(sh._lane, srs._h, 1, -d, dsc, N, C.c_void_p(parts.data_ptr()), 192) == 7
    assert L.sonic_msm_reduce_slices_dev(sh._lane, srs._h, C.c_void_p(full[0].data_ptr()), 1, S, 0, C.c_void_p(parts.data_ptr())) == 7
    assert L.sonic_msm_submit_dev(sh._lane, srs._h, 1, -d, dsc, N, C.c_void_p(parts.data_ptr())) == 7 and "retired" in _lib.last_error()
    assert L.sonic_msm_accumulate_dev(sh._lane, srs._h, 1, d - 5, dsc, N, C.c_void_p(full[0].data_ptr()), world * S) == 2      # leaves [-d, d]
    bad = sc.copy()
    bad[7, :] = 0xff
    dbad = _upload(L, _lib, bad)
    _lib.check(L.sonic_msm_accumulate_dev(sh._lane, srs._h, 1, -d, dbad, N, C.c_void_p(full[0].data_ptr()), world * S))
    assert L.sonic_msm_lane_sync(sh._lane) == 3
    sh.close()
    L.sonic_dev_free(dsc)
    L.sonic_dev_free(dbad)


def test_batch_of_64_proofs_streamed(sonic, orc):
    """BASELINE.json configs[4]: n = 2^16 (d = 8n = 2^19 because the reference rejects d < 7n), 64 transcripts"""
    n, Q = 1 << 16, 2
    d = 8 * n
    pyr = random.Random(64)
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    srs = sonic.SRS.new(d, x, alpha)
    circ = big_circuit(6464, n, Q)
    circuit = sonic.ArithCircuit(sonic.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
    asg = sonic.Assignment(circ["aL"], circ["aR"], circ["aO"])
    rng = np.random.default_rng(64)
    trs = [rand_fr_array(rng, 8 + 2 * Q) for _ in range(64)]
    for t in trs:
        t[:, 0] |= 1
    pipe = sonic.ProverPipeline(srs, circuit, depth=2)
    pipe.set_assignment(asg)
    streamed = pipe.prove_all(trs)
    assert len(streamed) == 64 and len(set(streamed)) == 64
    seq = sonic.Prover(srs, circuit)
    seq.set_assignment(asg)
    for i, t in enumerate(trs):
        assert seq.prove_bytes(t) == streamed[i], f"proof {i} differs between the pipeline and the sequential handle"
    o = orc.SRS(d, x, alpha, threads=NCPU)
    orc.set_mode(1, NCPU)
    for i in (0, 63):
        want = orc.prove(o, n, Q, circ["wL"], circ["wR"], circ["wO"], circ["cs"], circ["aL"], circ["aR"], circ["aO"], trs[i], True)
        assert streamed[i] == want, f"proof {i} differs from the C oracle"
    pipe.close()
    seq.close()


def test_host_threads_share_one_srs(sonic):
    """the reference's functions are pure and reentrant, so concurrent prove calls over ONE SRS handle from several
    host threads must work: three threads, each streaming 24 proofs (n = 2^12) through its own two-handle pipeline; four threads,
    each 20 MSMs on its own lane"""
    from sonic_amd import _lib
    L = _lib.lib()
    n, Q = 1 << 12, 2
    srs = sonic.SRS.new(8 * n, 0xabcdef123, 0x321fedcba)
    circ = big_circuit(12, n, Q)
    circuit = sonic.ArithCircuit(sonic.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
    asg = sonic.Assignment(circ["aL"], circ["aR"], circ["aO"])
    rng = np.random.default_rng(12)
    trs = [rand_fr_array(rng, 8 + 2 * Q) for _ in range(24)]
    for t in trs:
        t[:, 0] |= 1
    seq = sonic.Prover(srs, circuit)
    seq.set_assignment(asg)
    want = [seq.prove_bytes(t) for t in trs]
    res, errs = {}, []

    def work(k):
        try:
            pipe = sonic.ProverPipeline(srs, circuit, depth=2)
            pipe.set_assignment(asg)
            res[k] = pipe.prove_all(trs)
            pipe.close()
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert res[0] == want and res[1] == want and res[2] == want
    sc = rand_fr_array(rng, 50000)
    dp = _upload(L, _lib, sc)
    one = sonic.MsmLane()
    one.submit(srs, 0, -20000, dp, 50000)
    ref = one.collect()
    one.close()
    out = {}

    def mwork(k):
        try:
            ln = sonic.MsmLane()
            got = []
            for _ in range(20):
                ln.submit(srs, 0, -20000, dp, 50000)
                got.append(ln.collect())
            out[k] = got
            ln.close()
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=mwork, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert all(out[k] == [ref] * 20 for k in range(4))
    L.sonic_dev_free(dp)
    seq.close()


def test_readonly_calls_on_one_srs_from_many_threads(sonic, orc):
    """commitPoly / openPoly / the blocking MSM / hscProve on ONE SRS handle from 6 host threads at once (re-entrant calls on a
    shared SRS; round 4: each call leases its own stream + workspace instead of queueing behind a process-wide mutex): every result
    equals the one a single thread got, and the oracle's"""
    from sonic_amd.commitment import msm_g1_srs
    pyr = random.Random(4040)
    d = 1 << 12
    x, alpha = pyr.randrange(2, R), pyr.randrange(2, R)
    srs = sonic.SRS.new(d, x, alpha)
    osrs = orc.SRS(d, x, alpha, threads=NCPU)
    polys = []
    for k in range(6):
        exps = sorted(pyr.sample(range(-1500, 1500), 400 + 50 * k))
        exps = [e for e in exps if e != 0]
        polys.append(([(e, pyr.randrange(1, R)) for e in exps], pyr.randrange(1, R), 2000 + k))
    sc = rand_fr_array(np.random.default_rng(12), 3000)
    biv = {ex: {ey: pyr.randrange(1, R) for ey in range(-5, 6) if ey} for ex in range(-20, 21) if ex}
    yz = [(pyr.randrange(1, R), pyr.randrange(1, R))]

    def one(k):
        poly, z, mx = polys[k]
        return (sonic.commit_poly(srs, mx, poly), sonic.open_poly(srs, z, poly), msm_g1_srs(srs, k & 1, -1000 + k, sc),
                sonic.srs_points_bytes(srs, 0, -5 + k, 3) if hasattr(sonic, "srs_points_bytes") else srs.points(0, -5 + k, 3).tobytes(),
                sonic.hsc_prove_poly(srs, biv, yz, 17 + k, 19 + k))
    serial = [one(k) for k in range(6)]
    for k in range(6):                                       # the single-thread results against the oracle
        poly, z, mx = polys[k]
        exps = np.array([e for e, _ in poly], np.int64)
        co = fr_bytes([c for _, c in poly])
        assert sonic.g1_to_bytes(serial[k][0]) == orc.commit_poly(osrs, mx, exps, co)
        fz, W = orc.open_poly(osrs, z, exps, co)
        assert serial[k][1][0] == fz and sonic.g1_to_bytes(serial[k][1][1]) == W
        assert serial[k][2] == orc.msm_srs(osrs, k & 1, -1000 + k, sc, 1, NCPU)
    res, errs = {}, []

    def work(k):
        try:
            res[k] = [one(k) for _ in range(5)]
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for k in range(6):
        assert res[k] == [serial[k]] * 5, k


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


_RCCL_WORLD1 = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.distributed as dist
import sonic_amd
from sonic_amd import _lib, distributed as sd
from sonic_amd.workload import rand_fr_array
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
L = _lib.lib(); _lib.check(L.sonic_init(0))
d = 1 << 14
srs = sonic_amd.SRS.new(d, 0x1234567, 0x7654321)
N = 20000
sc = rand_fr_array(np.random.default_rng(1), N)
dsc = C.c_void_p(); _lib.check(L.sonic_dev_alloc(32 * N, C.byref(dsc))); _lib.check(L.sonic_dev_upload(dsc, sc.ctypes.data, 32 * N))
want = C.create_string_buffer(96); _lib.check(L.sonic_msm_g1_srs_dev(srs._h, 0, -10000, dsc, N, want))
sh = sd.ShardedMsm(srs, 0, 1, dev)
assert sh.pg and not sh.staged                    # device tensors over RCCL
assert sh.run_terms(0, -10000, dsc, N) == want.raw   # all_gather_into_tensor on the device
assert sh.run_buckets(0, -10000, dsc, N) == want.raw # all_to_all_single of the bucket ranges + all_gather_into_tensor
part = np.zeros(192, np.uint8)
_lib.check(L.sonic_msm_g1_srs_partial_dev(srs._h, 0, -10000, dsc, N, part.ctypes.data))
assert sd.sum_partials(sd.allgather_partials(part, 1, device=dev), 1) == want.raw
dist.barrier(); dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def test_rccl_process_group_of_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "-c", _RCCL_WORLD1], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_WORLD1_OK" in out.stdout, out.stderr[-3000:]


def test_bench_under_the_driver_launch_line_one_rank_rccl():
    """bench.py as the driver launches it for N > 1, with N = 1 and backend nccl: the process group exists, the MSM partials are
    all-gathered over RCCL on device tensors and the strong-scaling leg runs"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--log2n", "12", "--msm-log2", "14", "--no-cpu", "--strong-log2n", "13"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["config"]["process_group"].startswith("nccl") and j["value"] > 0
    assert j["msm"]["sequential"]["same_result_as_streamed"] and j["msm_strong"]["same_result_as_term_range_sharding"]
    assert j["roofline"]["bound"] == "hbm" and j["roofline_ntt"]["hbm_passes_per_transform"] >= 1 and j["roofline_prove"]["frac"] > 0
    # ONE proof shared by the (one) rank: the share went through the RCCL all-gather; the emulated 8-rank shares recombine to it
    ps = j["prove_strong"]
    assert ps["n"] == 1 << 13 and ps["n_gpus"] == 1 and ps["ms_per_proof"] > 0
    assert ps["emulated_shares"]["world"] == 8 and ps["emulated_shares"]["combined_equals_whole_proof"] is True
    assert j["north_star"]["n"] == 1 << 13 and j["north_star"]["ms_per_proof"] == ps["ms_per_proof"]
    # round 5: the reference's own call shape and the unprepared stream beside the headline, the C entry point of the throughput mode,
    # SURVEY 8d's second reading / sensitivities / protocol-shaped scalars, the executed-terms accounting, and a status the driver can read
    assert j["status"] == "ok" and j["leg_errors"] is None
    assert j["one_shot"]["same_bytes_as_streamed"] is True and j["one_shot"]["ms_per_proof"] > 0 and j["one_shot"]["first_call_ms"] > 0
    assert j["resident_unprepared"]["same_bytes_as_prepared"] is True and j["batch_c_abi"]["same_bytes_as_streamed"] is True
    n_, q_ = j["config"]["n"], j["config"]["Q"]
    rp = j["roofline_prove"]
    assert rp["scalar_muls_per_proof"] == 27 * n_ + 28 + 2 * q_ + q_ * (11 * n_ + q_) and rp["scalar_muls_executed_per_proof"] == 45 * n_ + 42
    assert j["resident_unprepared"]["scalar_muls_executed_per_proof"] == 49 * n_ + 40
    assert abs(rp["scalar_muls_per_s_inside_prove"] - rp["scalar_muls_executed_per_proof"] * j["value"]) / rp["scalar_muls_per_s_inside_prove"] < 1e-3
    sv = j["sensitivities"]
    assert sv["stated_d_reading"]["n"] == n_ // 2 and sv["stated_d_reading"]["d"] == 4 * n_ and sv["Q1"]["Q"] == 1 and sv["Q4"]["Q"] == 4
    assert sv["seed1"]["seed"] == 1 and sv["seed2"]["seed"] == 2 and all(v["ms_per_proof"] > 0 for k, v in sv.items() if k != "dense_weights")
    # round 6: what the headline owes to rndCircuit's all-ones rows (uniformly random weights: prepared stream, unprepared stream, one-shot
    # call -- all the same bytes), the integer roof of a whole proof, BASELINE configs[1] and configs[4] and the reference's own criterion shape
    dw = sv["dense_weights"]
    assert dw["n"] == n_ and dw["prepared_streamed"]["ms_per_proof"] > 0 and dw["resident_unprepared"]["same_bytes_as_prepared"] is True
    assert dw["one_shot"]["same_bytes_as_streamed"] is True and dw["one_shot"]["ms_per_proof"] > 0
    ip = j["int_roofline_prove"]
    assert ip["bound"] == "v_mad_u64_u32" and 0 < ip["frac"] < 1 and ip["plan"]["msms"] == 7 + 4 * q_
    c2, c5, cr = j["config2"], j["config5"], j["criterion_shape"]
    assert c2["n"] == 1 << 14 and c2["d"] == 1 << 17 and c2["same_bytes_streamed_and_sequential"] is True and c2["streamed"]["ms_per_proof"] > 0
    assert c5["n"] == 1 << 16 and c5["proofs"] == 64 and c5["same_bytes_as_one_handle_alone"] is True and c5["assignment_resident"]["proofs_per_s_per_gpu"] > 0
    assert cr["example_n1_Q2"]["verified"] is True and cr["example_n2_Q5"]["verified"] is True and cr["example_n2_Q5"]["d"] == 50
    ts = j["msm_strong"]["emulated_share"]["term_range_mode"]
    assert ts["ms_per_share"] > 0 and j["msm_strong"]["emulated_share"]["best_mode"] in ("bucket_ranges", "term_ranges")
    mp = j["msm_protocol_shaped"]
    assert mp["W_t_quotient"]["N"] == 7 * mp["n"] + 8 and mp["s_of_X_y_coefficients"]["N"] == 3 * mp["n"] + 1
    assert mp["s_of_X_y_coefficients"]["distinct_scalars"] <= mp["n"] + 3 and mp["W_t_quotient"]["distinct_scalars"] > 7 * mp["n"]
    e = j["msm_strong"]["emulated_share"]
    assert abs(e["ms_per_share"] - e["ms_per_share_kernels_only"] - e["exchange_model"]["ms"]) < 2e-3      # the modelled exchange is INSIDE the figure
    assert e["speedup_vs_single"] <= e["speedup_without_the_exchange"]
    assert ps["emulated_shares"]["allgather_model_ms"] > 0 and ps["emulated_shares"]["ranks_that_repeat_the_t_product"] >= 1


_NTT_ALT = r"""
import sys, hashlib
sys.path.insert(0, "tests")
import numpy as np
from sonic_amd import _lib
from util import rand_fr_array
L = _lib.lib(); _lib.check(L.sonic_init(0))
h = hashlib.sha256()
for log2n in (11, 13, 16):
    a = rand_fr_array(np.random.default_rng(log2n), 1 << log2n)
    for inverse in (0, 1):
        got = a.copy(); _lib.check(L.sonic_ntt_fr(got.ctypes.data, log2n, inverse)); h.update(got.tobytes())
g = np.random.default_rng(9)
pa, pb = rand_fr_array(g, 5000), rand_fr_array(g, 7000)
out = np.zeros((11999, 32), np.uint8)
_lib.check(L.sonic_poly_mul_fr(pa.ctypes.data, 5000, pb.ctypes.data, 7000, out.ctypes.data)); h.update(out.tobytes())
for na, nb in ((400000, 600000), (900000, 1100000)):       # products of 2^20 and 2^21 points: nine and ten wide stages
    pa, pb = rand_fr_array(g, na), rand_fr_array(g, nb)
    out = np.zeros((na + nb - 1, 32), np.uint8)
    _lib.check(L.sonic_poly_mul_fr(pa.ctypes.data, na, pb.ctypes.data, nb, out.ctypes.data)); h.update(out.tobytes())
a = rand_fr_array(np.random.default_rng(20), 1 << 20)
for inverse in (0, 1):
    got = a.copy(); _lib.check(L.sonic_ntt_fr(got.ctypes.data, 20, inverse)); h.update(got.tobytes())
print("NTT_DIGEST", h.hexdigest())
"""


def test_ntt_kernel_variants_agree():
    """the transform kernels exist in two builds of the same generated butterflies -- two per thread and four waves per SIMD (the
    default), four per thread and two waves (SONIC_NTT_WAVES=2) -- in any workgroup cap (SONIC_NTT_GRID), and with nine or ten wide stages
    in ONE pass through a 128-KB block (SONIC_NTT_BIG=1, round 5: faster alone on the chip, slower inside streamed proofs, so not the
    default): same bytes from all of them, up to transforms of 2^21 points (the default build is the one tests/test_gpu_parity.py and
    tests/test_gpu_fullsize.py hold against the oracle and the closed form)"""
    digests = []
    for extra in ({}, {"SONIC_NTT_WAVES": "2"}, {"SONIC_NTT_GRID": "3"}, {"SONIC_NTT_WAVES": "2", "SONIC_NTT_GRID": "5"}, {"SONIC_NTT_BIG": "1"}):
        out = subprocess.run([sys.executable, "-c", _NTT_ALT], cwd=ROOT, env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "NTT_DIGEST" in out.stdout, out.stderr[-3000:]
        digests.append(out.stdout.split("NTT_DIGEST")[1].split()[0])
    assert len(set(digests)) == 1, digests
