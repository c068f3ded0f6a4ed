#!/usr/bin/env python3
"""bench.py -- prove() proofs/s and G1 MSM scalar-muls/s on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: one full prove()
(Sonic.Protocol.prove incl. hscProve) on a synthetic random circuit (the reference's rndCircuit
generator at scale), circuit + assignment + SRS already resident in HBM when the timed region starts.
The K steps are streamed -- one host thread, two prover handles per GPU used in turn (sonic_prover_submit /
sonic_prover_collect), nothing synchronised between steps, all K proofs complete (and byte-identical to the
one-at-a-time proofs) at the closing barrier; `value` = K / that time.  The strictly sequential rate (each
prove() finished before the next begins = the latency of one proof) is reported beside it as `sequential`.
Workload (BASELINE.json configs[2], "n=2^18, d=2^20"): the reference rejects d < 7n
(src/Sonic/Protocol.hs:54-55), so prove() runs at the stated n = 2^18 with d = 8n = 2^21 and the
standalone MSM runs at exactly N = d = 2^20 terms (run A).  Q = 2.

N > 1: one process per GPU, SRS replicated.
  * prove() shards by proof (each rank proves its own proofs: no data-path collective)            -> `value`, weak scaling
  * `msm`: each rank one 2^20-term slice of an N*2^20-term MSM, 192-byte partials all-gathered    -> weak scaling
  * `msm_strong`: ONE fixed 2^22-term MSM (BASELINE.json configs[3]) split over the N ranks: term ranges for the
    accumulation, then an all-to-all of bucket ranges so that each rank reduces 1/N of the buckets, then the 192-byte
    gather                                                                                          -> strong scaling
All collectives run over RCCL on device tensors (sonic_amd/distributed.py); a process group of ONE rank (launched under
torch.distributed.run with N = 1) still executes them.

Every number of the roofline objects can be recomputed from the files they name under profiles/.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # one hardware queue per prover stream (see sonic_amd/__init__.py); before torch touches HIP

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ROUND = "r04"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_profile_json(name):
    """the newest profiles/rNN_<name> (this round's if it has been collected, else the previous round's), with its path"""
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{name}")))
    if not cands:
        return None, None
    try:
        return json.load(open(cands[-1])), os.path.relpath(cands[-1], ROOT)
    except Exception:
        return None, None


def rocprof_avg_ms(kernel):
    """average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of `bench.py --msm-only --msm-lanes 0`"""
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_msm_only_kernel_stats.csv")))
    if not cands:
        return None, None
    import csv
    for r in csv.DictReader(open(cands[-1])):
        if kernel in r.get("Name", ""):
            return float(r["AverageNs"]) / 1e6, os.path.relpath(cands[-1], ROOT)
    return None, os.path.relpath(cands[-1], ROOT)


def effective_cores():
    """host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=18, help="mult. gates n = 2^log2n (default: BASELINE configs[2])")
    ap.add_argument("--Q", type=int, default=2)
    ap.add_argument("--msm-log2", type=int, default=20)
    ap.add_argument("--msm-strong-log2", type=int, default=22, help="terms of the ONE MSM that is split over all ranks (BASELINE configs[3]); capped by the SRS")
    ap.add_argument("--msm-strong", action="store_true", help="only the strong-scaling MSM leg")
    ap.add_argument("--emulate-world", type=int, default=0, help="(1 GPU) time one rank's share of the strong-scaling MSM as if there were this many "
                                                                "ranks: 1/E of the terms, a device copy in place of the all-to-all, 1/E of the buckets")
    ap.add_argument("--cpu-log2n", type=int, default=0, help="n of the CPU-baseline proof (0: the bench's own n if a probe says it fits --cpu-budget-s, else the largest that does)")
    ap.add_argument("--cpu-budget-s", type=float, default=75.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print per-kernel HIP-event totals to stderr")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets several ranks share one GPU in tests)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the pipelined-throughput leg (timelines of one solo proof)")
    ap.add_argument("--msm-lanes", type=int, default=3, help="lanes the standalone MSMs are streamed over (0: skip the streamed leg, e.g. for rocprofv3 / PMC passes over the solo kernels)")
    ap.add_argument("--msm-only", action="store_true", help="skip prove() (PMC counter passes over the MSM kernels)")
    ap.add_argument("--strong-log2n", type=int, default=20, help="n = 2^this of the ONE proof that all ranks share (north_star: n = 2^20, d = 2^23; BASELINE configs[3] "
                                                                 "as a single instance); 0: skip the leg")
    ap.add_argument("--strong-emulate", type=int, default=8, help="(1 GPU) also time every rank's share of that proof as if there were this many ranks (0: off)")
    ap.add_argument("--north-star-cpu", action="store_true", help="also time the CPU port on the SAME n = 2^strong-log2n proof (minutes; one-off runs for profiles/)")
    ap.add_argument("--prove-only", action="store_true", help="nothing but the proofs (PMC pass for the per-kernel instruction budget of a proof)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # under torch.distributed.run: a process group even for N = 1
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_nccl = args.backend == "nccl"
    if launched:
        if use_nccl:
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)
    pg = launched
    coll_dev = device if use_nccl else torch.device("cpu")

    import sonic_amd
    from sonic_amd import _lib, distributed as sd
    from sonic_amd.workload import big_circuit, rand_fr_array
    L = _lib.lib()
    _lib.check(L.sonic_init(dev_index))

    n, Q = 1 << args.log2n, args.Q
    d = 8 * n
    msm_n = min(1 << args.msm_log2, 2 * d)          # the standalone MSM reads its points from this SRS (2d+1 per basis)
    strong_n = min(1 << args.msm_strong_log2, 2 * d)
    K, W = args.steps, args.warmup
    only_strong = args.msm_strong

    def barrier():
        if pg:
            dist.barrier()
        torch.cuda.synchronize()
        L.sonic_device_sync()

    def max_over_ranks(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=coll_dev)
        if pg:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    # ---------------- setup (untimed): SRS on the GPU, circuit resident in HBM ----------------
    t0 = time.time()
    seed_rng = np.random.default_rng(0)
    x = int.from_bytes(rand_fr_array(seed_rng, 1)[0].tobytes(), "little") | 1
    alpha = int.from_bytes(rand_fr_array(seed_rng, 1)[0].tobytes(), "little") | 1
    srs = sonic_amd.SRS.new(d, x, alpha)
    t_srs = time.time() - t0
    pipe = circ = None
    do_prove = not (args.msm_only or only_strong)
    depth = 1 if (args.kernel_table or args.no_pipeline) else 2
    if do_prove:       # --msm-only launches nothing but the stand-alone MSMs (so that a rocprofv3 summary of it is about them)
        circ = big_circuit(1000 + rank, n, Q)
        circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"])
        pipe = sonic_amd.ProverPipeline(srs, circuit, depth=depth)
        pipe.set_assignment(sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"]))
    tr_rng = np.random.default_rng(77 + rank)
    transcripts = [rand_fr_array(tr_rng, 8 + 2 * Q) for _ in range(K + W)]
    for t in transcripts:
        t[:, 0] |= 1                                   # evaluation points must be non-zero
    if rank == 0:
        log(f"setup: SRS.new(d=2^{args.log2n + 3}) {t_srs:.1f}s, circuit n=2^{args.log2n} Q={Q} resident")

    # ---------------- timed: K x prove() ----------------
    # The K proofs are streamed: one host thread, two prover handles used in turn (sonic_prover_submit / sonic_prover_collect), so
    # that proof i + 1 is already running while proof i is waited for and finished on the host -- nothing is synchronised between
    # steps, everything is complete at the closing barrier.  The strictly sequential number (every prove() call finished before
    # the next begins: the latency of one proof) is measured right after and reported beside it as "sequential".
    proof = b""
    dt_prove, proofs_per_s, sequential = 1.0, 0.0, None
    if do_prove:
        pipe.prove_all(transcripts[:max(W, depth)])           # warm-up (also grows every handle's workspaces)
        barrier()
        L.sonic_profile_reset()
        L.sonic_profile_enable(1 if args.kernel_table else 0)
        t0 = time.perf_counter()
        outs = pipe.prove_all(transcripts[W:W + K])
        barrier()
        dt = time.perf_counter() - t0
        L.sonic_profile_enable(0)
        proof = outs[-1] if outs else b""
        dt_prove = max_over_ranks([dt])[0]
        proofs_per_s = world * K / dt_prove
        if args.kernel_table and rank == 0:
            names = C.create_string_buffer(8192)
            L.sonic_profile_names(names, 8192)
            rows = []
            for nm in names.value.decode().split():
                ms, cnt = C.c_double(), C.c_int64()
                L.sonic_profile_get(nm.encode(), C.byref(ms), C.byref(cnt))
                rows.append((ms.value, cnt.value, nm))
            tot = sum(r[0] for r in rows)
            for ms, cnt, nm in sorted(rows, reverse=True):
                log(f"  {nm:28s} {ms:10.2f} ms {cnt:7d} launches {100 * ms / tot:5.1f}%")
            log(f"  kernels total {tot:.1f} ms of {dt * 1e3:.1f} ms wall ({K} proofs)")
        # strictly sequential (rank 0): the same K proofs, one finished prove() call after the other, on one handle
        if rank == 0 and K >= 1 and depth > 1:
            L.sonic_device_sync()
            t0 = time.perf_counter()
            for i in range(K):
                seq_proof = pipe.provers[0].prove_bytes(transcripts[W + i])
            L.sonic_device_sync()
            dts = time.perf_counter() - t0
            sequential = {"proofs_per_s_per_gpu": round(K / dts, 4), "ms_per_proof": round(1e3 * dts / K, 2),
                          "same_bytes_as_streamed": seq_proof == proof}
        for px in pipe.provers[1:]:
            px.close()
    barrier()

    # ---------------- NTT product alone on the chip (rank 0; roofline_ntt) ----------------
    ntt = None
    if do_prove and rank == 0 and not args.prove_only:
        na, nb = 3 * n + 5, 4 * n + 5                       # r(X,1) and r(X,y) + s(X,y): the shapes of tPoly's product (7n + 9 coefficients)
        lgM = (na + nb - 2).bit_length()
        M = 1 << lgM
        pa, pb = rand_fr_array(np.random.default_rng(5), na), rand_fr_array(np.random.default_rng(6), nb)
        da, db, do = C.c_void_p(), C.c_void_p(), C.c_void_p()
        for ptr, sz in ((da, 32 * na), (db, 32 * nb), (do, 32 * (na + nb - 1))):
            _lib.check(L.sonic_dev_alloc(sz, C.byref(ptr)))
        _lib.check(L.sonic_dev_upload(da, pa.ctypes.data, 32 * na))
        _lib.check(L.sonic_dev_upload(db, pb.ctypes.data, 32 * nb))
        for _ in range(2):
            _lib.check(L.sonic_poly_mul_fr_dev(da, na, db, nb, do))
        L.sonic_profile_reset()
        L.sonic_profile_enable(1)
        reps = 5
        for _ in range(reps):
            _lib.check(L.sonic_poly_mul_fr_dev(da, na, db, nb, do))
        L.sonic_profile_enable(0)
        per = {}
        for nm in ("k_ntt_wide", "k_ntt_local", "k_ntt_wide4", "k_ntt_local4", "k_fr_pointwise_mul"):       # (…4: the SONIC_NTT_WAVES=2 variants)
            ms, cnt = C.c_double(), C.c_int64()
            L.sonic_profile_get(nm.encode(), C.byref(ms), C.byref(cnt))
            per[nm] = {"ms_per_product": round(ms.value / reps, 4), "launches_per_product": cnt.value // reps}
        t_ms = sum(v["ms_per_product"] for v in per.values())
        pmc_ntt, pmc_ntt_src = load_profile_json("pmc_ntt.json")        # committed rocprofv3 --pmc passes of tools/ntt_time.py (per product, M = 2^21)
        passes = sum(per[k]["launches_per_product"] for k in ("k_ntt_wide", "k_ntt_local", "k_ntt_wide4", "k_ntt_local4")) // 3
        per = {k: v for k, v in per.items() if v["launches_per_product"] or k in ("k_ntt_wide", "k_ntt_local")}
        alg = 288.0 * M
        ntt = {"bound": "hbm", "kernel": "three radix-2 transforms of size M, the pointwise product folded into the inverse transform's first load (k_ntt_wide / k_ntt_local)",
               "M": M, "achieved": round(alg / (t_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(alg / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms_per_product": round(t_ms, 4),
               "algorithmic_bytes": alg, "algorithmic_bytes_rule": "288 M: each transform reads and writes M x 32 B once (3 x 64 M) + 96 M for the pointwise product (SURVEY 8d lower bound)",
               "hbm_passes_per_transform": passes, "bytes_by_design": float((3 * passes * 64 + 96) * M),
               "kernels": per, "measured": "HIP events around every launch of sonic_poly_mul_fr_dev, alone on the chip, 5 products",
               "traffic": (pmc_ntt or {}).get("hbm_bytes_per_product") if M == (pmc_ntt or {}).get("M") else None, "traffic_source": pmc_ntt_src,
               "rocprof_ms_per_product": (pmc_ntt or {}).get("rocprof_ms_per_product"),
               "note": "the HBM roof is SURVEY 8d's framing; with the wide stages fused five or six per pass through LDS (round 4) the transforms "
                       "are bound by VALU issue: ~356 instructions per butterfly in the generated assembly routines sonic_ntt_bfly2_fwd / _inv (DESIGN.md section 5)"}
        for ptr in (da, db, do):
            L.sonic_dev_free(ptr)

    # ---------------- timed: standalone G1 MSM, N = 2^20 per GPU, scalars resident in HBM ----------------
    sc = rand_fr_array(np.random.default_rng(500), max(msm_n, strong_n))       # the same scalars on every rank (the strong leg splits them)
    dsc = C.c_void_p()
    _lib.check(L.sonic_dev_alloc(32 * sc.shape[0], C.byref(dsc)))
    _lib.check(L.sonic_dev_upload(dsc, sc.ctypes.data, 32 * sc.shape[0]))
    msm = roofline = int_roofline = None
    accum_ms, kern_total = 0.0, 0.0
    if not only_strong and not args.prove_only:
        basis, e0 = sd.msm_shard(rank, world, d, msm_n)
        # (1) one MSM after the other, every launch bracketed by HIP events: the dominant kernel's duration for the roofline
        #     (alone on the chip, as in the rocprofv3 summary of --msm-only) and the latency of one MSM.  The partial stays on
        #     the device, the all-gather runs over RCCL on the lane's stream, one device-to-host copy brings the gathered partials.
        one = sd.ShardedMsm(srs, rank, world, device)
        for _ in range(W):
            one.run_terms(basis, e0, dsc, msm_n)
        barrier()
        L.sonic_profile_reset()
        L.sonic_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(K):
            seq_result = one.run_terms(basis, e0, dsc, msm_n)
        barrier()
        dt_seq = time.perf_counter() - t0
        L.sonic_profile_enable(0)
        one.close()
        # (2) the same K MSMs streamed over NL lanes: MSM i + 1 is queued before MSM i is collected, so the sort and the
        #     latency-bound reduction of one run under the accumulation of the other; each MSM's partial still goes through the
        #     all-gather and the curve additions.  `msm.value` is this throughput.
        NL = max(0, args.msm_lanes)
        lanes = [sonic_amd.MsmLane() for _ in range(NL)]

        def msm_stream(count):
            res = b""
            if count <= 0:
                return res
            for j in range(min(NL - 1, count)):
                lanes[j % NL].submit(srs, basis, e0, dsc, msm_n)
            for i in range(count):
                if i + NL - 1 < count:
                    lanes[(i + NL - 1) % NL].submit(srs, basis, e0, dsc, msm_n)
                mine = np.frombuffer(lanes[i % NL].collect(partial=True), np.uint8)
                parts = sd.allgather_partials(mine, world, device=coll_dev if use_nccl else None)
                res = sd.sum_partials(parts, world)
            return res

        if NL > 0:
            msm_stream(max(W, NL))
            barrier()
            t0 = time.perf_counter()
            stream_result = msm_stream(K)
            barrier()
            dt = time.perf_counter() - t0
        else:
            stream_result, dt = seq_result, dt_seq
        dt_msm, dt_msm_seq = max_over_ranks([dt, dt_seq])
        msm_per_s = world * msm_n * K / dt_msm
        for ln in lanes:
            ln.close()
        ms, cnt = C.c_double(), C.c_int64()
        L.sonic_profile_get(b"k_bucket_accum", C.byref(ms), C.byref(cnt))
        accum_ms = ms.value / max(1, cnt.value)
        names = C.create_string_buffer(8192)
        L.sonic_profile_names(names, 8192)
        per_kernel = {}
        for nm in names.value.decode().split():
            m2, c2 = C.c_double(), C.c_int64()
            L.sonic_profile_get(nm.encode(), C.byref(m2), C.byref(c2))
            kern_total += m2.value
            per_kernel[nm] = round(m2.value / max(1, K), 4)
            if args.kernel_table and rank == 0:
                log(f"  [msm] {nm:24s} {m2.value / max(1, c2.value):9.3f} ms/launch x{c2.value}")
        msm = {"metric": "G1 MSM scalar-muls/sec", "value": round(msm_per_s, 1), "unit": "scalar-muls/s", "N_per_gpu": msm_n, "scaling": "weak",
               "ms_per_msm": round(1e3 * dt_msm / K, 3),
               "streaming": (f"K MSMs streamed over {NL} lanes per GPU (submit / collect); one at a time in `sequential`" if NL > 0 else "none (--msm-lanes 0): one MSM at a time"),
               "sequential": {"scalar_muls_per_s": round(world * msm_n * K / dt_msm_seq, 1), "ms_per_msm": round(1e3 * dt_msm_seq / K, 3),
                              "kernel_ms_per_msm": round(kern_total / K, 3), "kernel_ms": per_kernel, "same_result_as_streamed": seq_result == stream_result}}

    # Secondary legs must not take the headline down with them: nothing below has run on more than one GPU before the driver's own
    # multi-GPU run, so a leg that raises is recorded as {"error": ...} (every rank takes the same path through a leg, so a rank that
    # fails before a collective fails on all ranks alike; a one-sided failure would surface as the collective's timeout).
    leg_errors = {}

    # ---------------- timed: ONE 2^22-term MSM split over all ranks (strong scaling; BASELINE configs[3]) ----------------
    msm_strong = None
    if not args.msm_only and not args.prove_only:
        try:
            sh = sd.ShardedMsm(srs, rank, world, device)
            lo, hi = sd.split_range(strong_n, world, rank)
            dmine = C.c_void_p(dsc.value + 32 * lo)
            e_lo = -d + lo
            try:
                sd.exchange_layout(srs, world)
                exchange = world > 1
            except _lib.SonicError:            # an SRS without window tables (SONIC_MSM_TABLES=0, or d too large for them): term ranges only
                exchange = False
            if exchange:
                run = lambda: sh.run_buckets(0, e_lo, dmine, hi - lo)             # noqa: E731
            elif world > 1:
                run = lambda: sh.run_terms(0, e_lo, dmine, hi - lo)               # noqa: E731
            else:
                run = lambda: sh.run_terms(0, -d, dsc, strong_n)                  # noqa: E731  (one rank: the plain MSM is the baseline of the curve)
            for _ in range(max(1, W)):
                res_strong = run()
            check = sh.run_terms(0, e_lo, dmine, hi - lo)                          # the same sum by term-range partials (every rank its slice)
            barrier()
            t0 = time.perf_counter()
            for _ in range(K):
                res_strong = run()
            barrier()
            dt_strong = max_over_ranks([time.perf_counter() - t0])[0]
            msm_strong = {"metric": "one G1 MSM split over all ranks", "N_total": strong_n, "scaling": "strong", "n_gpus": world,
                          "ms_per_msm": round(1e3 * dt_strong / K, 3), "value": round(strong_n * K / dt_strong, 1), "unit": "scalar-muls/s",
                          "method": ("term ranges accumulated per rank, all-to-all of bucket ranges (RCCL), 1/N of the buckets reduced per rank, 192-B all-gather"
                                     if exchange else ("term ranges, 192-B all-gather (no window tables on this SRS)" if world > 1 else
                                                       "single rank: the plain MSM (baseline of the strong-scaling curve)")),
                          "same_result_as_term_range_sharding": res_strong == check}
            if args.emulate_world > 1 and world == 1:
                E = args.emulate_world
                lo_e, hi_e = sd.split_range(strong_n, E, 0)
                for _ in range(max(1, W)):
                    sh.run_buckets_emulated(0, -d, dsc, hi_e - lo_e, E)
                L.sonic_device_sync()
                L.sonic_profile_reset()
                L.sonic_profile_enable(1)
                t0 = time.perf_counter()
                for _ in range(K):
                    sh.run_buckets_emulated(0, -d, dsc, hi_e - lo_e, E)
                L.sonic_device_sync()
                dte = time.perf_counter() - t0
                L.sonic_profile_enable(0)
                names = C.create_string_buffer(8192)
                L.sonic_profile_names(names, 8192)
                perk = {}
                for nm in names.value.decode().split():
                    m2, c2 = C.c_double(), C.c_int64()
                    L.sonic_profile_get(nm.encode(), C.byref(m2), C.byref(c2))
                    perk[nm] = round(m2.value / K, 4)
                # the exchange the emulation leaves out, as a modelled term: each rank sends one slice to each of the E - 1 peers, every pair on
                # its own xGMI link (point-to-point, 7 links x ~153 GB/s per GPU: MI355X_MICROARCH.md), so the all-to-all takes one slice over
                # one link; 70 % of the link rate assumed attainable + 20 us for the collective's launch and synchronisation
                _, S_e = sd.exchange_layout(srs, E)
                slice_bytes = S_e * 192
                xch_ms = 1e3 * slice_bytes / (0.7 * 153e9) + 0.02
                share_ms = 1e3 * dte / K
                msm_strong["emulated_share"] = {"world": E, "terms": hi_e - lo_e, "ms_per_share": round(share_ms, 3), "kernel_ms": perk,
                                                "speedup_vs_single": round((dt_strong / K) / (dte / K), 2),
                                                "exchange_model": {"bytes_per_pair": slice_bytes, "link_GBps": 153, "assumed_efficiency": 0.7, "fixed_ms": 0.02,
                                                                   "ms": round(xch_ms, 3)},
                                                "ms_per_share_with_modelled_exchange": round(share_ms + xch_ms, 3),
                                                "speedup_with_modelled_exchange": round(1e3 * (dt_strong / K) / (share_ms + xch_ms), 2),
                                                "note": "UNMEASURED ON MULTI-GPU HARDWARE: one GPU doing one rank's work, device copy instead of the xGMI all-to-all "
                                                        "(its time is the modelled term, not a measurement)"}
            sh.close()
        except Exception as e:      # noqa: BLE001
            leg_errors["msm_strong"] = repr(e)
            msm_strong = {"error": repr(e)}
    L.sonic_dev_free(dsc)

    # ---------------- timed: ONE proof at the north_star size shared by all ranks (strong scaling of prove()) ----------------
    # Every rank holds the same circuit, assignment and transcript over its replica of the SRS, runs its cost-balanced piece of the
    # proof's 7 + 4Q MSMs (sonic_prover_set_share) and the ranks all-gather their shares (a few KB).  One rank: the plain sequential
    # prove() -- the north_star's "prove() wall-clock at n = 2^20 on 1 MI355X".
    prove_strong = north_star = None
    if do_prove and not args.prove_only and args.strong_log2n > 0:
        try:
            ns_lg = args.strong_log2n
            ns_n, ns_d = 1 << ns_lg, 8 << ns_lg
            t0 = time.time()
            srs_ns = srs if ns_d == d else sonic_amd.SRS.new(ns_d, x, alpha)
            t_srs_ns = time.time() - t0
            c_ns = circ if (ns_n == n and world == 1) else big_circuit(2000, ns_n, Q)           # the same statement on every rank
            circuit_ns = sonic_amd.ArithCircuit(sonic_amd.GateWeights(c_ns["wL"], c_ns["wR"], c_ns["wO"]), c_ns["cs"])
            asg_ns = sonic_amd.Assignment(c_ns["aL"], c_ns["aR"], c_ns["aO"])
            sp = sd.ShardedProver(srs_ns, circuit_ns, rank, world, device)
            sp.set_assignment(asg_ns)
            ns_rng = np.random.default_rng(4242)
            ns_tr = [rand_fr_array(ns_rng, 8 + 2 * Q) for _ in range(K + max(1, W))]
            for t in ns_tr:
                t[:, 0] |= 1
            for i in range(max(1, W)):
                sp.prove_bytes(ns_tr[i])
            barrier()
            t0 = time.perf_counter()
            for i in range(K):
                ns_proof = sp.prove_bytes(ns_tr[max(1, W) + i])
            barrier()
            dt_ns = max_over_ranks([time.perf_counter() - t0])[0]
            # what every rank spends on its own share (no collective in the timed part): the slowest and the mean over the ranks show how
            # well the plan balances on real hardware, next to the end-to-end time above
            share_stats = None
            if world > 1:
                t_sh = []
                for _ in range(3):
                    L.sonic_device_sync()
                    t0 = time.perf_counter()
                    sp.prove_share(ns_tr[max(1, W) + K - 1])
                    t_sh.append(time.perf_counter() - t0)
                mine_ms = 1e3 * min(t_sh)
                t = torch.tensor([mine_ms], dtype=torch.float64, device=coll_dev)
                tmax, tsum = t.clone(), t.clone()
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
                share_stats = {"slowest_share_ms": round(float(tmax.item()), 3), "mean_share_ms": round(float(tsum.item()) / world, 3)}
            same_ns = None
            if rank == 0 and world > 1:       # the same proof made by this GPU alone (untimed): the bytes must not depend on the sharing
                alone = sonic_amd.Prover(srs_ns, circuit_ns, prepare=False)
                alone.set_assignment(asg_ns)
                same_ns = alone.prove_bytes(ns_tr[max(1, W) + K - 1]) == ns_proof
                alone.close()
            prove_strong = {"metric": "ONE prove() shared by all ranks", "n": ns_n, "Q": Q, "d": ns_d, "scaling": "strong", "n_gpus": world,
                            "ms_per_proof": round(1e3 * dt_ns / K, 3), "value": round(K / dt_ns, 4), "unit": "proofs/s",
                            "method": ("every rank builds the polynomials its pieces read and runs a contiguous, cost-balanced piece of the proof's 7+4Q MSMs "
                                       "(cuts inside an MSM split its term range); one all-gather of %d-byte shares; sonic_proof_from_shares on every rank"
                                       % L.sonic_proof_share_size(Q)) if world > 1 else "single rank: the plain sequential prove() (baseline of the curve)",
                            "same_bytes_as_one_gpu_alone": same_ns, "proof_bytes": len(ns_proof), "srs_new_s": round(t_srs_ns, 2), "shares": share_stats}
            if world == 1 and args.strong_emulate > 1:
                E = args.strong_emulate
                tr_e = ns_tr[max(1, W) + K - 1]
                ms_e, shares_e = [], []
                for r in range(E):
                    sp.set_emulated_rank(r, E)
                    sp.prove_share(tr_e)
                    L.sonic_device_sync()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        sh_e = sp.prove_share(tr_e)
                    ms_e.append(1e3 * (time.perf_counter() - t0) / 3)
                    shares_e.append(sh_e)
                t0 = time.perf_counter()
                comb = sonic_amd.proof_from_shares(Q, shares_e, tr_e)
                t_comb = 1e3 * (time.perf_counter() - t0)
                prove_strong["emulated_shares"] = {"world": E, "ms_per_share": [round(v, 2) for v in ms_e], "slowest_ms": round(max(ms_e), 2),
                                                   "combine_ms_host": round(t_comb, 3), "combined_equals_whole_proof": comb == ns_proof,
                                                   "speedup_vs_one_gpu": round((1e3 * dt_ns / K) / (max(ms_e) + t_comb), 2),
                                                   "note": "UNMEASURED ON MULTI-GPU HARDWARE: this one GPU ran every rank's share in turn; "
                                                           "the all-gather of %d bytes per rank over xGMI is not included" % L.sonic_proof_share_size(Q)}
            if rank == 0 and world == 1:
                north_star = {"target": "prove() wall-clock at n=2^20 (d = 8n = 2^23; BASELINE states d=2^22, which Protocol.hs:54-55 rejects) on 1 MI355X, "
                                        ">= 10x the CPU prove(), bit-exact", "n": ns_n, "d": ns_d, "Q": Q,
                              "ms_per_proof": round(1e3 * dt_ns / K, 3), "how": f"{K} sequential prove() calls, each finished before the next begins"}
                # the CPU port on this very proof takes ~55 s: not part of the default run (it would double it); the committed measurement
                # of `bench.py --north-star-cpu` on a box of the same pool is quoted beside the live GPU time, marked as such
                ref_ns, ref_src = load_profile_json("north_star.json")
                if ref_ns and ref_ns.get("north_star", {}).get("cpu") and ref_ns["north_star"].get("n") == ns_n:
                    c_ref = ref_ns["north_star"]["cpu"]
                    north_star["cpu_from_profile"] = {"file": ref_src, "s_per_proof": c_ref["s_per_proof"], "cores": c_ref["cores"], "kind": c_ref["kind"],
                                                      "same_bytes_as_gpu_proof_in_that_run": c_ref["same_bytes_as_gpu_proof"],
                                                      "gpu_ms_in_that_run": ref_ns["north_star"]["ms_per_proof"],
                                                      "ratio_vs_this_run": round(c_ref["s_per_proof"] / (dt_ns / K), 1),
                                                      "note": "NOT timed in this run: `python bench.py --north-star-cpu` times it live"}
                if args.north_star_cpu and not args.no_cpu:
                    from oracle import orc
                    cores_ns = effective_cores()
                    orc.set_mode(1, cores_ns)
                    o_ns = orc.SRS.from_points(ns_d, srs_ns.points(0, -ns_d, 2 * ns_d + 1), srs_ns.points(1, -ns_d, 2 * ns_d + 1))
                    t0 = time.perf_counter()
                    cp_ns = orc.prove(o_ns, ns_n, Q, c_ns["wL"], c_ns["wR"], c_ns["wO"], c_ns["cs"], c_ns["aL"], c_ns["aR"], c_ns["aO"], ns_tr[max(1, W) + K - 1], True)
                    cdt_ns = time.perf_counter() - t0
                    north_star["cpu"] = {"kind": "port", "what": "oracle/sonic_oracle.c, the repo's plain-C port (Pippenger + NTT)", "cores": cores_ns, "n": ns_n,
                                         "s_per_proof": round(cdt_ns, 2), "same_bytes_as_gpu_proof": cp_ns == ns_proof,
                                         "gpu_over_cpu": round(cdt_ns / (dt_ns / K), 1)}
                    del o_ns
            sp.close()
            if srs_ns is not srs:
                del srs_ns

        except Exception as e:      # noqa: BLE001
            leg_errors["prove_strong"] = repr(e)
            prove_strong = {"error": repr(e)}
            north_star = None

    if rank != 0:
        if pg:
            dist.destroy_process_group()
        return

    if msm is not None:
        # roofline of the dominant kernel (k_bucket_accum of the N = 2^20 MSM): algorithmic bytes = 128 B per
        # scalar-mul (96 B affine point + 32 B scalar) x the terms one launch covers
        alg_bytes = 128.0 * msm_n
        achieved = alg_bytes / (accum_ms * 1e-3) / 1e9 if accum_ms > 0 else 0.0
        pmc, pmc_src = load_profile_json("pmc_msm.json")
        traffic = pmc["k_bucket_accum"]["hbm_bytes_per_launch"] if pmc and pmc.get("msm_n") == msm_n and "k_bucket_accum" in pmc else None
        rp_ms, rp_src = rocprof_avg_ms("k_bucket_accum")
        roofline = {"bound": "hbm", "kernel": "k_bucket_accum", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_uses": "avg_launch_ms (live HIP events of THIS run)",
                    "traffic": traffic, "traffic_source": pmc_src if traffic is not None else None,
                    "avg_launch_ms": round(accum_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
                    "rocprof": {"avg_launch_ms": None if rp_ms is None else round(rp_ms, 4), "summary": rp_src,
                                "frac": None if not rp_ms else round(alg_bytes / (rp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                "note": "rocprofv3 --kernel-trace --stats of `bench.py --msm-only --msm-lanes 0 --no-cpu --steps 5 --warmup 1` on another box of the pool "
                                        "(same launches, nothing else); run-to-run and box-to-box spread of this kernel is a few per cent"},
                    "note": "modular-integer kernel: the binding roof is integer multiply issue, see int_roofline"}
        # integer roof of the same kernel: additions x MADs per addition against the chip's v_mad_u64_u32 issue rate; counts and
        # rates are read from profiles/rNN_kernel_model.json (tools/kernel_model.py: static ISA count of the compiled kernel +
        # the on-hardware microbenchmarks), not constants in this file.
        pc, pw_, pb = C.c_int(), C.c_int(), C.c_int()
        L.sonic_msm_plan(srs._h, msm_n, C.byref(pc), C.byref(pw_), C.byref(pb))
        # by design the table method reads a 4-B sorted entry and one table point (one 128-B line since the tables are padded) per (term, window)
        roofline["bytes_by_design_per_launch"] = float((4 + L.sonic_srs_point_bytes()) * pw_.value * msm_n) if pb.value == 1 else None
        model, model_src = load_profile_json("kernel_model.json")
        n_adds = pw_.value * msm_n - (1 << (pc.value - 1)) * pb.value
        adds_per_s = n_adds / (accum_ms * 1e-3) if accum_ms > 0 else 0.0
        if model:
            mads, instr = model["mads_per_addition"], model["instr_per_addition"]
            int_roofline = {"bound": "v_mad_u64_u32", "achieved": round(adds_per_s * mads / 1e12, 3), "peak": model["mad_peak_per_s"] / 1e12, "unit": "TMAD/s",
                            "frac": round(adds_per_s * mads / model["mad_peak_per_s"], 4),
                            "plan": {"window_bits": pc.value, "windows": pw_.value, "bucket_sets": pb.value},
                            "additions_per_launch": n_adds, "instr_per_addition": instr, "mads_per_addition": mads,
                            "alu_only": {"adds_per_s_register_loop": model.get("addition_register_loop_per_s"), "kernel_adds_per_s": round(adds_per_s, 1),
                                         "frac": round(adds_per_s / model["addition_register_loop_per_s"], 4) if model.get("addition_register_loop_per_s") else None},
                            "model": model_src}
        else:
            int_roofline = {"note": "profiles/rNN_kernel_model.json missing (tools/kernel_model.py)", "additions_per_launch": n_adds, "kernel_adds_per_s": round(adds_per_s, 1)}

    # whole prove(): SURVEY 8d's algorithmic bytes against the time of one streamed proof
    roofline_prove = None
    if do_prove:
        lgM = (7 * n + 8).bit_length()
        scalar_muls = 27 * n + 28 + 2 * Q + Q * (11 * n + Q)
        alg_p = 128.0 * scalar_muls + 288.0 * (1 << lgM)
        roofline_prove = {"bound": "hbm", "algorithmic_bytes_per_proof": alg_p, "scalar_muls_per_proof": scalar_muls, "ntt_size": 1 << lgM,
                          "rule": "128 B x (27n + 28 + 2Q + Q(11n + Q)) + 288 M (SURVEY 8d)", "achieved": round(alg_p * proofs_per_s / world / 1e9, 2),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_p * proofs_per_s / world / 1e9 / HBM_PEAK_GBS, 5),
                          "scalar_muls_per_s_inside_prove": round(scalar_muls * proofs_per_s, 1)}

    cpu_baseline = None
    if not args.no_cpu and do_prove:
        from oracle import orc    # the CPU oracle is only ever the baseline leg here, never part of the GPU path
        cores = effective_cores()
        cr = np.random.default_rng(3)
        orc.set_mode(1, cores)

        def cpu_prove_time(lg, osrs_, cc, tr_):
            m = 1 << lg
            t0_ = time.perf_counter()
            pb_ = orc.prove(osrs_, m, Q, cc["wL"], cc["wR"], cc["wO"], cc["cs"], cc["aL"], cc["aR"], cc["aO"], tr_, True)
            return time.perf_counter() - t0_, pb_

        # probe at n = 2^12 on an oracle-made SRS (cost is ~linear in n), then ONE proof at the largest n <= the bench's n that
        # fits the budget -- on the bench's own SRS, circuit and transcript when that is the bench's n, so that the CPU proof
        # can be compared with the GPU's byte for byte.  No extrapolation: what is printed was timed.
        plg = min(12, args.log2n)
        pcirc = big_circuit(1, 1 << plg, Q)
        ptr = rand_fr_array(cr, 8 + 2 * Q)
        ptr[:, 0] |= 1
        psrs = orc.SRS(8 << plg, x, alpha, threads=cores)
        probe, _ = cpu_prove_time(plg, psrs, pcirc, ptr)
        cpu_lg = args.cpu_log2n if args.cpu_log2n > 0 else args.log2n
        if args.cpu_log2n <= 0:
            while cpu_lg > plg and probe * (1 << (cpu_lg - plg)) > args.cpu_budget_s:
                cpu_lg -= 1
        same = None
        if cpu_lg == args.log2n:
            # the SRS is set-up, not part of prove(): the oracle takes the GPU-made points (orc_srs_from_points) instead of
            # spending minutes of fixed-base multiplications on the host
            t0 = time.perf_counter()
            osrs = orc.SRS.from_points(d, srs.points(0, -d, 2 * d + 1), srs.points(1, -d, 2 * d + 1))
            t_osrs = time.perf_counter() - t0
            cdt, cproof = cpu_prove_time(cpu_lg, osrs, circ, transcripts[W + K - 1])
            same = cproof == proof
            srs_note = f"SRS points taken from the GPU-made SRS ({t_osrs:.1f}s copy, untimed)"
        else:
            cn_ = 1 << cpu_lg
            t0 = time.perf_counter()
            osrs = orc.SRS(8 * cn_, x, alpha, threads=cores)
            t_osrs = time.perf_counter() - t0
            ccirc = big_circuit(1, cn_, Q)
            cdt, _ = cpu_prove_time(cpu_lg, osrs, ccirc, ptr)
            srs_note = f"oracle-made SRS ({t_osrs:.1f}s, untimed)"
        cmsm_n = min(msm_n, 2 * (8 << cpu_lg))
        t0 = time.perf_counter()
        orc.msm_srs(osrs, 0, -(8 << cpu_lg), sc[:cmsm_n], 1, cores)
        cmsm_dt = time.perf_counter() - t0
        # the reference-shaped cost ("cpu-literal"): per-term double-and-add fold for the MSMs
        # (CommitmentScheme.hs:26-29) and the schoolbook product for tPoly, one thread as the reference never forks; small n only
        orc.set_mode(0, 1)
        lit_n = 256
        lc = big_circuit(1, lit_n, Q)
        lsrs = orc.SRS(8 * lit_n, x, alpha, threads=cores)
        t0 = time.perf_counter()
        orc.prove(lsrs, lit_n, Q, lc["wL"], lc["wR"], lc["wO"], lc["cs"], lc["aL"], lc["aR"], lc["aO"], ptr, False)
        lit_dt = time.perf_counter() - t0
        orc.set_mode(1, cores)
        cpu_baseline = {"value": round(1.0 / cdt, 5), "unit": "proofs/s", "cores": cores, "kind": "port",
                        "n": 1 << cpu_lg, "at_bench_size": cpu_lg == args.log2n, "s_per_proof": round(cdt, 2), "same_bytes_as_gpu_proof": same,
                        "sample": f"oracle/sonic_oracle.c (Pippenger + NTT, {cores} threads = usable host cores of {os.cpu_count()} visible) ONE prove() at n=2^{cpu_lg}, Q={Q}, d=8n: "
                                  f"{cdt:.2f}s; {srs_note}; sized by a {probe:.2f}s probe at n=2^{plg} against a {args.cpu_budget_s:.0f}s budget",
                        "msm_scalar_muls_per_s": round(cmsm_n / cmsm_dt, 1), "msm_sample": f"ONE N={cmsm_n} Pippenger MSM, {cores} threads, {cmsm_dt:.2f}s",
                        "literal": {"n": lit_n, "s_per_proof": round(lit_dt, 2), "cores": 1,
                                    "note": "the oracle with the reference's algorithms (fold of per-term double-and-add, schoolbook tPoly): "
                                            "cost grows like n^2 in tPoly and 380 group operations per term in the MSMs"}}

    line = {
        "metric": "prove() proofs/sec",
        "value": round(proofs_per_s, 4),
        "unit": "proofs/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(1e3 * dt_prove / K, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 limbs (Fq 12x32, Fr 8x32 Montgomery)",
        "data": "synthetic",
        "config": {"workload": f"prove(): rndCircuit n=2^{args.log2n}, Q={Q}, SRS d=2^{args.log2n + 3} (d=8n >= 7n, Protocol.hs:54); "
                               f"G1 MSM N=2^{args.msm_log2} per GPU; one G1 MSM N=2^{strong_n.bit_length() - 1} over all GPUs", "n": n, "Q": Q, "d": d,
                   "sharding": "proof-per-rank (value); MSM term-range-sharded (msm, weak); one MSM bucket-range-sharded (msm_strong, strong)",
                   "streaming": "K proofs streamed by one host thread through 2 prover handles per GPU (submit / collect); "
                                "the strictly sequential rate is in `sequential`",
                   "process_group": (f"{args.backend}, {world} rank(s)" if pg else "none (plain single-process run)")},
        "msm": msm,
        "msm_strong": msm_strong,
        "roofline": roofline,
        "int_roofline": int_roofline,
        "roofline_ntt": ntt,
        "roofline_prove": roofline_prove,
        "cpu_baseline": cpu_baseline,
        "prove_strong": prove_strong,
        "north_star": north_star,
        "proof_bytes": len(proof),
        "sequential": sequential,
        "leg_errors": leg_errors or None,
    }
    print(json.dumps(line), flush=True)
    if pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
