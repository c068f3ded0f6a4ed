"""N GPUs from ONE host process, through the C ABI alone (round 5).

The reference's prove is one pure call in one process (src/Sonic/Protocol.hs:47-52).  sonic_prove_shared / sonic_prove_batch /
sonic_msm_g1_srs_multi keep it that way on a node of GPUs: handles carry their device, the library runs one host thread per handle.
tests/host/multi_harness.c drives them from plain C99 with a device LIST -- [0, 0, 0] on the pool's one-GPU boxes: three handles
sharing the GPU run exactly the code three GPUs run (threads, shares, peer copies of bucket ranges); a >= 2-GPU box also runs the
list of real ordinals.  Expected bytes: the CPU oracle's proof (python restatement fixtures for the golden cases, the C oracle
beyond), so "shared over N handles" == "one GPU" == "oracle"."""
import json
import os
import random
import struct
import subprocess

import numpy as np
import pytest

from util import R, big_circuit, circuit_arrays, fr_bytes, rand_fr_array

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "prove_small.json")))["cases"]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("mh") / "multi_harness")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(HERE, "host", "multi_harness.c"), "-L" + os.path.join(ROOT, "sonic_amd", "csrc"), "-lsonic_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "sonic_amd", "csrc"), "-o", exe])
    return exe


def _run(exe, case_path, devices):
    env = dict(os.environ, SONIC_TORCH_PRELOAD="0")
    out = subprocess.run([exe, str(case_path), ",".join(str(v) for v in devices)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "multi_harness: OK" in out.stdout, out.stdout + out.stderr[-3000:]
    return out.stdout


def _case_blob(n, Q, d, x, alpha, enc, transcripts, want):
    return struct.pack("<qqqq", n, Q, d, len(transcripts)) + fr_bytes([x]).tobytes() + fr_bytes([alpha]).tobytes() + \
        b"".join(np.ascontiguousarray(enc[k], np.uint8).tobytes() for k in ("wL", "wR", "wO", "cs", "aL", "aR", "aO")) + \
        b"".join(np.ascontiguousarray(t, np.uint8).tobytes() for t in transcripts) + want


def _device_lists():
    import torch
    lists = [[0, 0, 0]]
    nd = torch.cuda.device_count()
    if nd >= 2:
        lists.append(list(range(min(nd, 8))))
    return lists


@pytest.mark.parametrize("case", [c for c in GOLD if c["name"] in ("example1", "example2", "rnd_n3", "rnd_n8")], ids=lambda c: c["name"])
def test_c99_multi_harness_on_the_golden_cases(sonic, harness, tmp_path, case):
    """the committed fixtures (Example1 / Example2 of test/Test/Reference.hs:38-90, rndCircuit cases) through one process driving a
    list of devices: expected bytes = the fixture's proof (python restatement)"""
    c = case
    iv = lambda v: int(v, 16)          # noqa: E731
    flat = lambda w: fr_bytes([iv(v) for r in w for v in r])    # noqa: E731
    enc = dict(wL=flat(c["wL"]), wR=flat(c["wR"]), wO=flat(c["wO"]), cs=fr_bytes([iv(v) for v in c["cs"]]),
               aL=fr_bytes([iv(v) for v in c["aL"]]), aR=fr_bytes([iv(v) for v in c["aR"]]), aO=fr_bytes([iv(v) for v in c["aO"]]))
    pyr = random.Random(c["n"])
    trs = [fr_bytes([iv(v) for v in c["transcript"]])] + [fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * c["Q"])]) for _ in range(4)]
    path = tmp_path / "case.bin"
    path.write_bytes(_case_blob(c["n"], c["Q"], c["d"], iv(c["x"]), iv(c["alpha"]), enc, trs, bytes.fromhex(c["proof"])))
    for devs in _device_lists():
        _run(harness, path, devs)


@pytest.mark.parametrize("n,Q", [(16, 2), (1000, 3), (1 << 14, 2)])
def test_c99_multi_harness_against_the_c_oracle(sonic, orc, ref, harness, tmp_path, n, Q):
    """n in {16, 1000, 2^14}: expected bytes = the C oracle's proof over the GPU-made SRS points"""
    pyr = random.Random(5000 + n)
    d = 8 * n
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    if n <= 64:
        _, _, enc = circuit_arrays(ref, pyr, n, Q)
    else:
        enc = big_circuit(n + Q, n, Q)
    K = 5
    trs = [fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)]) for _ in range(K)]
    srs = sonic.SRS.new(d, x, alpha)
    osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1))
    orc.set_mode(1, os.cpu_count() or 1)
    want = orc.prove(osrs, n, Q, enc["wL"], enc["wR"], enc["wO"], enc["cs"], enc["aL"], enc["aR"], enc["aO"], trs[0], True)
    srs.close()
    path = tmp_path / "case.bin"
    path.write_bytes(_case_blob(n, Q, d, x, alpha, enc, trs, want))
    for devs in _device_lists():
        _run(harness, path, devs)


def test_python_mirror_of_the_multi_device_entry_points(sonic, orc):
    """sonic_amd.prove_shared / prove_batch / msm_g1_srs_multi (ctypes over the same symbols) with replicas on the device list
    [0, 0] -- and on real ordinals where the box has them; SRS.device, SRS.replicate, device_count"""
    import torch
    nd = sonic.device_count()
    assert nd == torch.cuda.device_count() and nd >= 1
    devs = [0, 0] if nd < 2 else [0, 1]
    n, Q = 600, 2
    d = 8 * n
    pyr = random.Random(99)
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    enc = big_circuit(7, n, Q)
    circuit = sonic.ArithCircuit(sonic.GateWeights(enc["wL"], enc["wR"], enc["wO"]), enc["cs"])
    asg = sonic.Assignment(enc["aL"], enc["aR"], enc["aO"])
    reps = [sonic.SRS.new(d, x, alpha, device=devs[0])]
    reps.append(reps[0].replicate(devs[1]))
    assert [r.device for r in reps] == devs
    assert np.array_equal(reps[0].points(1, -5, 11), reps[1].points(1, -5, 11))
    provers = [sonic.Prover(r, circuit, prepare=(i == 0)) for i, r in enumerate(reps)]      # mixed on purpose: shares must agree ...
    for p in provers:
        p.set_assignment(asg)
    trs = [rand_fr_array(np.random.default_rng(40 + i), 8 + 2 * Q) for i in range(6)]
    for t in trs:
        t[:, 0] |= 1
    alone = sonic.Prover(reps[0], circuit, prepare=False)
    alone.set_assignment(asg)
    want = [alone.prove_bytes(t) for t in trs]
    # ... a prepared and an unprepared handle plan differently: the combine says so (plan tag) instead of producing a wrong proof
    from sonic_amd import _lib
    with pytest.raises(_lib.SonicError) as e:
        sonic.prove_shared(provers, trs[0])
    assert e.value.code == 7 and "different parameters" in e.value.message
    provers[1].close()
    provers[1] = sonic.Prover(reps[1], circuit, prepare=True)
    provers[1].set_assignment(asg)
    assert sonic.prove_shared(provers, trs[0]) == want[0]
    for p in provers:
        p.set_share(0, 1)
    assert sonic.prove_batch(provers, trs) == want
    assert sonic.prove_batch(provers, trs[:3], [asg] * 3) == want[:3]
    assert sonic.prove_batch(provers, []) == []
    # one MSM over the two replicas
    sc = rand_fr_array(np.random.default_rng(3), 5000)
    from sonic_amd.commitment import msm_g1_srs
    one = msm_g1_srs(reps[0], 0, -2000, sc)
    assert sonic.msm_g1_srs_multi(reps, 0, -2000, sc, mode=0) == one
    assert sonic.msm_g1_srs_multi(reps, 0, -2000, sc, mode=1) == one
    osrs = orc.SRS.from_points(d, reps[0].points(0, -d, 2 * d + 1), reps[0].points(1, -d, 2 * d + 1))
    assert orc.msm_srs(osrs, 0, -2000, sc, 1, os.cpu_count() or 1) == one
    for p in provers + [alone]:
        p.close()


def test_prove_many_independent_statements(sonic, orc, ref):
    """BASELINE configs[4] read literally -- a batch of INDEPENDENT proofs: every statement its own circuit (dense weights, the rndCircuit
    rows, and one with an all-zero constraint row), assignment and transcript, handed over as host buffers (`mapM (uncurry (prove srs))`,
    src/Sonic/Protocol.hs:47-52) -- through sonic_prove_many over the replica list [0, 0] (real ordinals where the box has them): bytes
    against the C oracle for every statement; a non-canonical assignment in one statement is that statement's status and the others are
    still proven; the one-shot shells parked on the device are re-used across calls and shapes"""
    from sonic_amd import _lib
    nd = sonic.device_count()
    devs = [0, 0] if nd < 2 else [0, 1]
    n, Q = 700, 3
    d = 7 * n + 11
    pyr = random.Random(77)
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    reps = [sonic.SRS.new(d, x, alpha, device=devs[0])]
    reps.append(reps[0].replicate(devs[1]))
    osrs = orc.SRS.from_points(d, reps[0].points(0, -d, 2 * d + 1), reps[0].points(1, -d, 2 * d + 1))
    orc.set_mode(1, os.cpu_count() or 1)
    statements, want = [], []
    for i in range(11):
        c = big_circuit(300 + i, n, Q)
        if i % 3 == 1:                       # dense random weights
            la, lb, lo = c["ints"]
            rng = np.random.default_rng(i)
            W = [rand_fr_array(rng, Q * n) for _ in range(3)]
            if i == 4:
                for w in W:
                    w[n:2 * n] = 0           # an all-zero constraint row
            cs = []
            for q in range(Q):
                acc = 0
                for w, a in zip(W, (la, lb, lo)):
                    acc += sum(int.from_bytes(w[q * n + k].tobytes(), "little") * a[k] for k in range(n))
                cs.append(acc % R)
            c = dict(c, wL=W[0], wR=W[1], wO=W[2], cs=fr_bytes(cs))
        tr = fr_bytes([pyr.randrange(1, R) for _ in range(8 + 2 * Q)])
        statements.append((sonic.Assignment(c["aL"], c["aR"], c["aO"]), sonic.ArithCircuit(sonic.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"]), tr))
        want.append(orc.prove(osrs, n, Q, c["wL"], c["wR"], c["wO"], c["cs"], c["aL"], c["aR"], c["aO"], tr, True))
    assert sonic.prove_many(reps, statements) == want
    assert sonic.prove_many(reps[:1], statements[:3]) == want[:3]            # one replica: two threads on it
    assert sonic.prove_many(reps, []) == []
    # a bad statement in the middle: its status is the call's, the proofs around it are made
    bad = list(statements)
    a5 = bad[5][0]
    broken = np.array(a5.aR, copy=True)
    broken[3, :] = 0xFF
    bad[5] = (sonic.Assignment(a5.aL, broken, a5.aO), bad[5][1], bad[5][2])
    with pytest.raises(_lib.SonicError) as e:
        sonic.prove_many(reps, bad)
    assert e.value.code == 3 and "statement 5" in e.value.message
    # another shape afterwards (the parked shells of the first shape make room), then the first shape again
    n2 = 90
    c2 = big_circuit(9, n2, 1)
    st2 = (sonic.Assignment(c2["aL"], c2["aR"], c2["aO"]), sonic.ArithCircuit(sonic.GateWeights(c2["wL"], c2["wR"], c2["wO"]), c2["cs"]),
           fr_bytes([pyr.randrange(1, R) for _ in range(10)]))
    got2 = sonic.prove_many(reps, [st2] * 5)
    assert len(set(got2)) == 1 and got2[0] == orc.prove(osrs, n2, 1, c2["wL"], c2["wR"], c2["wO"], c2["cs"], c2["aL"], c2["aR"], c2["aO"], st2[2], True)
    assert sonic.prove_many(reps, statements[:4]) == want[:4]
    # a shape the SRS is too small for (Protocol.hs:54-55) is reported per statement
    big = big_circuit(1, d // 7 + 1, 1)
    with pytest.raises(_lib.SonicError) as e:
        sonic.prove_many(reps, [(sonic.Assignment(big["aL"], big["aR"], big["aO"]), sonic.ArithCircuit(sonic.GateWeights(big["wL"], big["wR"], big["wO"]), big["cs"]),
                                 fr_bytes([pyr.randrange(1, R) for _ in range(10)]))])
    assert e.value.code == 1


def test_srs_pairing_record_field(sonic):
    """srsPairing = e(g, h^alpha) (src/Sonic/SRS.hs:21,42) from the GPU-made G2 half through the host pairing, against the python
    oracle's pairing of the same two points (tests/test_sanitizers.py pins the same entry point GPU-free)"""
    from test_sanitizers import srs_pairing_bytes
    pyr = random.Random(17)
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    srs = sonic.SRS.new(8, x, alpha)
    want = srs_pairing_bytes(alpha)
    v = [int.from_bytes(want[48 * i:48 * i + 48], "little") for i in range(12)]
    assert srs.srsPairing == tuple(tuple((v[6 * i + 2 * j], v[6 * i + 2 * j + 1]) for j in range(3)) for i in range(2))
    # a handle built from G1 points only has no G2 half: the field cannot be served
    g1only = sonic.SRS.from_points(8, srs.points(0, -8, 17), srs.points(1, -8, 17))
    from sonic_amd import _lib
    with pytest.raises(_lib.SonicError) as e:
        g1only.srsPairing
    assert e.value.code == 7


def test_concurrent_multi_device_calls_from_several_host_threads(sonic):
    """the in-process entry points are re-entrant: two host threads run sonic_prove_shared over their own pairs of handles while a third
    runs one MSM over two replicas by bucket range and a fourth streams a batch -- all on the device list [0, 0] (or real ordinals) --
    and every result equals the single-handle one (per-device contexts, pooled lanes and leased streams instead of process-wide state)"""
    import threading
    nd = sonic.device_count()
    devs = [0, 0] if nd < 2 else [0, 1]
    n, Q = 900, 2
    d = 8 * n
    pyr = random.Random(4)
    x, alpha = pyr.randrange(1, R), pyr.randrange(1, R)
    enc = big_circuit(11, n, Q)
    circuit = sonic.ArithCircuit(sonic.GateWeights(enc["wL"], enc["wR"], enc["wO"]), enc["cs"])
    asg = sonic.Assignment(enc["aL"], enc["aR"], enc["aO"])
    reps = [sonic.SRS.new(d, x, alpha, device=devs[0])]
    reps.append(reps[0].replicate(devs[1]))

    def handles():
        hs = [sonic.Prover(r, circuit, prepare=True) for r in reps]
        for h in hs:
            h.set_assignment(asg)
        return hs
    groups = [handles() for _ in range(3)]
    trs = [rand_fr_array(np.random.default_rng(70 + i), 8 + 2 * Q) for i in range(8)]
    for t in trs:
        t[:, 0] |= 1
    alone = sonic.Prover(reps[0], circuit, prepare=False)
    alone.set_assignment(asg)
    want = [alone.prove_bytes(t) for t in trs]
    sc = rand_fr_array(np.random.default_rng(5), 6000)
    from sonic_amd.commitment import msm_g1_srs
    want_msm = msm_g1_srs(reps[0], 1, -3000, sc[:2999])
    errors, results = [], {}

    def shared(g, which):
        try:
            for rnd in range(4):
                for i in which:
                    assert sonic.prove_shared(groups[g], trs[i]) == want[i], (g, i)
            results[("shared", g)] = True
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    def msm():
        try:
            for rnd in range(6):
                assert sonic.msm_g1_srs_multi(reps, 1, -3000, sc[:2999], mode=rnd & 1) == want_msm
            results["msm"] = True
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    def batch():
        try:
            for rnd in range(3):
                assert sonic.prove_batch(groups[2], trs) == want
            results["batch"] = True
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=shared, args=(0, [0, 1, 2, 3])), threading.Thread(target=shared, args=(1, [4, 5, 6, 7])),
          threading.Thread(target=msm), threading.Thread(target=batch)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(results) == 4
    for g in groups:
        for h in g:
            h.close()
    alone.close()
