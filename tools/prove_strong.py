#!/usr/bin/env python3
"""One GPU standing in, in turn, for every rank of a proof that is shared over E GPUs (sonic_prover_set_share): the time of each
rank's share against the whole proof on the same GPU.  The slowest share + the all-gather of a few KB is what one proof would take
on E GPUs; nothing here has run on more than one GPU (profiles/rNN_prove_strong_emulated.txt says so too).

  python tools/prove_strong.py --log2n 20 --worlds 2,4,8 --steps 3 [--fit]

--fit: least-squares fit of the plan's cost model (share_plan.hpp: ms per term, per MSM piece, per polynomial) to the measured
share times, printed as the SONIC_SHARE_COST_* values it implies.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

NAMES = None


def slot_names(Q):
    nm = {0: "R", 1: "T", 2: "Wa", 3: "Wb", 4: "Wt", 5 + 4 * Q: "Qv", 6 + 4 * Q: "C"}
    for j in range(Q):
        nm[5 + 2 * j] = f"S{j + 1}"; nm[6 + 2 * j] = f"W{j + 1}"; nm[5 + 2 * Q + 2 * j] = f"W'{j + 1}"; nm[6 + 2 * Q + 2 * j] = f"Q{j + 1}"
    return nm


def slot_terms(n, Q, prepared):
    t = {0: 3 * n + 4, 1: 7 * n + 9, 2: 3 * n + 4, 3: 3 * n + 4, 4: 7 * n + 8, 5 + 4 * Q: 2 * n + Q, 6 + 4 * Q: 2 * n + Q + 1}
    for j in range(Q):
        t[5 + 2 * j] = n if prepared else 3 * n + 1; t[6 + 2 * j] = 3 * n; t[5 + 2 * Q + 2 * j] = 3 * n; t[6 + 2 * Q + 2 * j] = 2 * n + Q
    return t


def features(pieces, n, Q, prepared):
    """(terms, pieces, r1, sy, su, tprod) of a rank's share, as share_plan.hpp counts them"""
    T = slot_terms(n, Q, prepared)
    terms = sum(T[i] * (hi - lo) / (1 << 20) for i, (lo, hi) in enumerate(pieces) if hi > lo)
    jobs = sum(1 for lo, hi in pieces if hi > lo)
    own = lambda i: pieces[i][1] > pieces[i][0]                                   # noqa: E731
    need_T = own(1) or own(4)
    r1 = int(need_T or own(0) or own(2) or own(3))
    sy = int(need_T) + sum(int(own(5 + 2 * j) or own(6 + 2 * j) or own(5 + 2 * Q + 2 * j)) for j in range(Q))
    su = int(own(6 + 4 * Q) or own(5 + 4 * Q) or any(own(6 + 2 * Q + 2 * j) for j in range(Q)))
    return [terms, jobs, r1, sy, su, int(need_T)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--Q", type=int, default=2)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--fit", action="store_true")
    ap.add_argument("--no-prepare", action="store_true")
    args = ap.parse_args()
    import sonic_amd
    from sonic_amd import _lib
    from sonic_amd.workload import big_circuit, rand_fr_array
    import ctypes as C
    L = _lib.lib()
    _lib.check(L.sonic_init(0))
    n, Q = 1 << args.log2n, args.Q
    d = 8 * n
    prepared = not args.no_prepare
    t0 = time.time()
    srs = sonic_amd.SRS.new(d, 0x1234567891, 0x9876543211)
    circ = big_circuit(1000, n, Q)
    circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
    p = sonic_amd.Prover(srs, circuit, prepare=prepared)
    p.set_assignment(sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"]))
    tr = rand_fr_array(np.random.default_rng(77), 8 + 2 * Q)
    tr[:, 0] |= 1
    c, w, sets = C.c_int(), C.c_int(), C.c_int()
    _lib.check(L.sonic_msm_plan(srs._h, 3 * n, C.byref(c), C.byref(w), C.byref(sets)))
    nb = 1 << (c.value - 1)
    print(f"# prove() n=2^{args.log2n} Q={Q} d=2^{args.log2n + 3} prepared={prepared}; MSM plan c={c.value} W={w.value} bucket sets={sets.value}; "
          f"setup {time.time() - t0:.1f}s; one MI355X standing in for every rank in turn (UNMEASURED ON MULTI-GPU HARDWARE)")

    def timed(fn):
        fn(); fn()
        L.sonic_device_sync()
        t = time.perf_counter()
        for _ in range(args.steps):
            out = fn()
        L.sonic_device_sync()
        return 1e3 * (time.perf_counter() - t) / args.steps, out

    whole_ms, whole = timed(lambda: p.prove_bytes(tr))
    print(f"whole proof on one GPU (one at a time): {whole_ms:.2f} ms")
    nm = slot_names(Q)
    rows, times = [], []
    for E in [int(v) for v in args.worlds.split(",")]:
        plan = sonic_amd.share_plan(n, Q, prepared, E, nb=nb, w=w.value)
        shares, ms = [], []
        for r in range(E):
            p.set_share(r, E)
            t_ms, sh = timed(lambda: p.prove_share(tr))
            shares.append(sh); ms.append(t_ms)
            rows.append(features(plan[r][0], n, Q, prepared)); times.append(t_ms)
        same = sonic_amd.proof_from_shares(Q, shares, tr) == whole
        t_comb = time.perf_counter()
        for _ in range(20):
            sonic_amd.proof_from_shares(Q, shares, tr)
        t_comb = 1e3 * (time.perf_counter() - t_comb) / 20
        print(f"E = {E}: slowest share {max(ms):.2f} ms + combine {t_comb:.3f} ms on the host -> {whole_ms / (max(ms) + t_comb):.2f}x one GPU "
              f"(mean share {sum(ms) / E:.2f} ms; combined proof == whole proof: {same})")
        for r in range(E):
            pcs = " ".join(f"{nm[i]}[{lo / (1 << 20):.2f},{hi / (1 << 20):.2f})" if (lo, hi) != (0, 1 << 20) else nm[i]
                           for i, (lo, hi) in enumerate(plan[r][0]) if hi > lo)
            print(f"    rank {r}: {ms[r]:7.2f} ms  model {plan[r][1] / n:5.2f} n   {pcs}")
    p.set_share(0, 1)
    if args.fit and len(rows) >= 7:
        A = np.array(rows, float)
        A[:, 0] /= n
        A = np.hstack([A, np.ones((len(rows), 1))])
        x, *_ = np.linalg.lstsq(A, np.array(times), rcond=None)
        per_n = x[0]
        print(f"# fit: {per_n:.3f} ms per n terms, {x[1]:.3f} ms per MSM piece, r1 {x[2]:.3f}, s(X,y) {x[3]:.3f}, s(u,Y) {x[4]:.3f}, t product {x[5]:.3f}, constant {x[6]:.3f} ms")
        print(f"# -> SONIC_SHARE_COST_JOB={x[1] / per_n * n * w.value / nb:.2f} SONIC_SHARE_COST_R1={x[2] / per_n:.2f} SONIC_SHARE_COST_SY={x[3] / per_n:.2f} "
              f"SONIC_SHARE_COST_SU={x[4] / per_n:.2f} SONIC_SHARE_COST_T={x[5] / per_n:.2f}  (residual rms {np.sqrt(np.mean((A @ x - np.array(times)) ** 2)):.2f} ms)")


if __name__ == "__main__":
    main()
