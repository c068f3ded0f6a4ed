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

N > 1: one process per GPU.  prove() shards by proof (each rank proves its own proofs on a replicated
SRS: no data-path collective); the standalone MSM is range-sharded (each rank one 2^20-term slice of
an N*2^20-term MSM) and the 192-byte partial sums are all-gathered over RCCL, then added on every rank.
Weak scaling in both cases.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # one hardware queue per prover stream (see sonic_amd/__init__.py); before torch touches HIP

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Integer roof of the bucket accumulation (tools/microbench.hip on one MI355X, profiles/r02_microbench.txt):
MAD_PEAK_PER_S = 2.8e13        #   v_mad_u64_u32 issue rate, all CUs, independent chains
MADD_ALU_ONLY_PER_S = 6.51e9   #   the same fused mixed addition in a register-only loop (no memory access): what the instruction mix sustains
# static ISA count of one bucket-walk iteration (tools/count_accum_instrs.py): 816 instructions around six calls of the 624-instruction
# product core, two of the 532-instruction squaring core and one of the 924-instruction two-product core
ACCUM_INSTR_PER_ADD = 6548
ACCUM_MADS_PER_ADD = 2608
PMC_FILE = os.path.join(ROOT, "profiles", "r02_pmc_msm.json")   # FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=18, help="mult. gates n = 2^log2n (default: BASELINE configs[2])")
    ap.add_argument("--Q", type=int, default=2)
    ap.add_argument("--msm-log2", type=int, default=20)
    ap.add_argument("--cpu-log2n", type=int, default=0, help="n of the bounded CPU-baseline sample (0: sized from a probe at n=2^11 "
                                                              "so that one CPU proof takes <= ~5 s, at most n=2^15)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print per-kernel HIP-event totals to stderr")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo lets several ranks share one GPU in tests)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the pipelined-throughput leg (timelines of one solo proof)")
    ap.add_argument("--msm-lanes", type=int, default=3, help="lanes the standalone MSMs are streamed over (0: skip the streamed leg, e.g. for rocprofv3 / PMC passes over the solo kernels)")
    ap.add_argument("--msm-only", action="store_true", help="skip prove() (PMC counter passes over the MSM kernels)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    use_nccl = args.backend == "nccl"
    if world > 1:
        if use_nccl:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)
    coll_dev = torch.device("cuda", dev_index) if use_nccl else torch.device("cpu")

    import sonic_amd
    from sonic_amd import _lib
    from util import big_circuit, rand_fr_array, R
    L = _lib.lib()
    _lib.check(L.sonic_init(dev_index))

    n, Q = 1 << args.log2n, args.Q
    d = 8 * n
    msm_n = min(1 << args.msm_log2, 2 * d)          # the standalone MSM reads its points from this SRS (2d+1 per basis)
    K, W = args.steps, args.warmup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        L.sonic_device_sync()

    # ---------------- setup (untimed): SRS on the GPU, circuit resident in HBM ----------------
    t0 = time.time()
    seed_rng = np.random.default_rng(0)
    x = int.from_bytes(rand_fr_array(seed_rng, 1)[0].tobytes(), "little") | 1
    alpha = int.from_bytes(rand_fr_array(seed_rng, 1)[0].tobytes(), "little") | 1
    srs = sonic_amd.SRS.new(d, x, alpha)
    t_srs = time.time() - t0
    prover = None
    if not args.msm_only:       # --msm-only launches nothing but the stand-alone MSMs (so that a rocprofv3 summary of it is about them)
        circ = big_circuit(1000 + rank, n, Q, None)
        prover = sonic_amd.Prover(srs, sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"]))
        prover.set_assignment(sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"]))
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
    if args.msm_only:
        K_prove, W_prove = 0, 0
    else:
        K_prove, W_prove = K, W
    proof = b""
    dt_prove, proofs_per_s, sequential, pipe = 1.0, 0.0, None, None
    if prover is not None:
        depth = 1 if (args.kernel_table or args.no_pipeline) else 2
        pipe = sonic_amd.ProverPipeline.__new__(sonic_amd.ProverPipeline)
        pipe.provers = [prover]
        for _ in range(depth - 1):
            px = sonic_amd.Prover(srs, sonic_amd.ArithCircuit(sonic_amd.GateWeights(circ["wL"], circ["wR"], circ["wO"]), circ["cs"]))
            px.set_assignment(sonic_amd.Assignment(circ["aL"], circ["aR"], circ["aO"]))
            pipe.provers.append(px)
        pipe.prove_all(transcripts[:max(W_prove, depth)])           # warm-up (also grows every handle's workspaces)
        barrier()
        L.sonic_profile_reset()
        L.sonic_profile_enable(1 if args.kernel_table else 0)
        t0 = time.perf_counter()
        outs = pipe.prove_all(transcripts[W:W + K_prove])
        barrier()
        dt = time.perf_counter() - t0
        L.sonic_profile_enable(0)
        proof = outs[-1] if outs else b""
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_prove = float(tmax.item())
        proofs_per_s = world * K_prove / dt_prove
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
        if rank == 0 and K_prove >= 1 and depth > 1:
            L.sonic_device_sync()
            t0 = time.perf_counter()
            for i in range(K_prove):
                seq_proof = prover.prove_bytes(transcripts[W + i])
            L.sonic_device_sync()
            dts = time.perf_counter() - t0
            sequential = {"proofs_per_s_per_gpu": round(K_prove / dts, 4), "ms_per_proof": round(1e3 * dts / K_prove, 2),
                          "same_bytes_as_streamed": seq_proof == proof}
        for px in pipe.provers[1:]:
            px.close()
    barrier()

    # ---------------- timed: standalone G1 MSM, N = 2^20 per GPU, scalars resident in HBM ----------------
    sc = rand_fr_array(np.random.default_rng(500 + rank), msm_n)
    dsc = C.c_void_p()
    _lib.check(L.sonic_dev_alloc(32 * msm_n, C.byref(dsc)))
    _lib.check(L.sonic_dev_upload(dsc, sc.ctypes.data, 32 * msm_n))
    from sonic_amd import distributed as sd
    basis, e0 = sd.msm_shard(rank, world, d, msm_n)
    part = np.zeros(sd.PARTIAL_BYTES, np.uint8)
    msm_result = [b""]

    def msm_step():
        _lib.check(L.sonic_msm_g1_srs_partial_dev(srs._h, basis, e0, dsc, msm_n, part.ctypes.data))
        parts = sd.allgather_partials(part, world, device=coll_dev if use_nccl else None)       # RCCL all-gather of 192 B
        msm_result[0] = sd.sum_partials(parts, world)                                            # k-1 curve additions

    # (1) one MSM after the other, every launch bracketed by HIP events: the dominant kernel's duration for the roofline
    #     (alone on the chip, as in the rocprofv3 summary of --msm-only) and the latency of one MSM
    for _ in range(W):
        msm_step()
    barrier()
    L.sonic_profile_reset()
    L.sonic_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(K):
        msm_step()
    barrier()
    dt_seq = time.perf_counter() - t0
    L.sonic_profile_enable(0)
    seq_result = msm_result[0]
    # (2) the same K MSMs streamed over two lanes (sonic_msm_submit / sonic_msm_collect): MSM i + 1 is queued before MSM i is
    #     collected, so the sort and the latency-bound reduction of one run under the accumulation of the other; each MSM's
    #     partial still goes through the all-gather and the curve additions.  `msm.value` is this throughput.
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
    tmax = torch.tensor([dt, dt_seq], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_msm, dt_msm_seq = float(tmax[0].item()), float(tmax[1].item())
    msm_per_s = world * msm_n * K / dt_msm
    for ln in lanes:
        ln.close()
    ms, cnt = C.c_double(), C.c_int64()
    L.sonic_profile_get(b"k_bucket_accum", C.byref(ms), C.byref(cnt))
    accum_ms = ms.value / max(1, cnt.value)
    kern_total = 0.0
    names = C.create_string_buffer(8192)
    L.sonic_profile_names(names, 8192)
    for nm in names.value.decode().split():
        m2, c2 = C.c_double(), C.c_int64()
        L.sonic_profile_get(nm.encode(), C.byref(m2), C.byref(c2))
        kern_total += m2.value
        if args.kernel_table and rank == 0:
            log(f"  [msm] {nm:24s} {m2.value / max(1, c2.value):9.3f} ms/launch x{c2.value}")
    L.sonic_dev_free(dsc)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # roofline of the dominant kernel (k_bucket_accum of the N = 2^20 MSM): algorithmic bytes = 128 B per
    # scalar-mul (96 B affine point + 32 B scalar) x the terms one launch covers
    alg_bytes = 128.0 * msm_n
    achieved = alg_bytes / (accum_ms * 1e-3) / 1e9 if accum_ms > 0 else 0.0
    traffic, traffic_src = None, None
    try:   # HBM bytes per launch from the separate PMC passes (same command with --msm-only), if taken for this size
        pmc = json.load(open(PMC_FILE))
        if pmc.get("msm_n") == msm_n:
            traffic, traffic_src = pmc["k_bucket_accum"]["hbm_bytes_per_launch"], "profiles/r02_pmc_msm.json"
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "k_bucket_accum", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(accum_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_by_design_per_launch": None,   # filled below: the window-table method reads one 96-B table point per (term, window)
                "rocprof_summary": "profiles/r02_msm_only_kernel_stats.csv (rocprofv3 --kernel-trace --stats -- python3 bench.py --msm-only --msm-lanes 0 --no-cpu "
                                   "--steps 5 --warmup 1: the same N = 2^20 launches and nothing else; profiles/r02_bench_kernel_stats.csv is the full "
                                   "default run, where the kernel also serves the batched groups of prove())",
                "note": "modular-integer kernel: the binding roof is integer multiply issue, see int_roofline"}
    # integer roof of the same kernel.  Additions = entries - buckets (the first entry of a bucket is a copy); every addition is
    # ACCUM_MADS_PER_ADD v_mad_u64_u32 (the multiplier: ~2.3x the issue cost of a plain 32-bit VALU instruction on this chip) inside
    # ACCUM_INSTR_PER_ADD instructions.  Two fractions: of the chip's MAD issue rate, and of the rate the whole instruction mix
    # sustains in a register-only loop (what is left is memory latency and the launch tail).
    pc, pw_, pb = C.c_int(), C.c_int(), C.c_int()
    L.sonic_msm_plan(srs._h, msm_n, C.byref(pc), C.byref(pw_), C.byref(pb))
    roofline["bytes_by_design_per_launch"] = float((4 + 96) * pw_.value * msm_n) if pb.value == 1 else None   # 4-B sorted entry + 96-B table point per (term, window)
    n_adds = pw_.value * msm_n - (1 << (pc.value - 1)) * pb.value
    adds_per_s = n_adds / (accum_ms * 1e-3) if accum_ms > 0 else 0.0
    int_roofline = {"bound": "v_mad_u64_u32", "achieved": round(adds_per_s * ACCUM_MADS_PER_ADD / 1e12, 3),
                    "peak": MAD_PEAK_PER_S / 1e12, "unit": "TMAD/s", "frac": round(adds_per_s * ACCUM_MADS_PER_ADD / MAD_PEAK_PER_S, 4),
                    "plan": {"window_bits": pc.value, "windows": pw_.value, "bucket_sets": pb.value},
                    "mixed_additions_per_launch": n_adds, "instr_per_mixed_add": ACCUM_INSTR_PER_ADD, "mads_per_mixed_add": ACCUM_MADS_PER_ADD,
                    "alu_only": {"adds_per_s_register_loop": MADD_ALU_ONLY_PER_S, "kernel_adds_per_s": round(adds_per_s, 1),
                                 "frac": round(adds_per_s / MADD_ALU_ONLY_PER_S, 4),
                                 "source": "tools/microbench g1_add_mixed_walk (profiles/r02_microbench.txt)"},
                    "pmc": "profiles/r02_pmc_SQ_counter_collection.csv: SQ_INSTS_VALU per launch against n_adds x instr_per_mixed_add / 64"}

    cpu_baseline = None
    if not args.no_cpu:
        from oracle import orc    # the CPU oracle is only ever the baseline leg here, never part of the GPU path
        cores = os.cpu_count() or 1
        cr = np.random.default_rng(3)
        cx = int.from_bytes(rand_fr_array(cr, 1)[0].tobytes(), "little") | 1
        ca = int.from_bytes(rand_fr_array(cr, 1)[0].tobytes(), "little") | 1
        ctr = rand_fr_array(cr, 8 + 2 * Q)
        ctr[:, 0] |= 1
        orc.set_mode(1, cores)

        def cpu_prove_time(lg, budget_s, max_reps):
            m = 1 << lg
            osrs_ = orc.SRS(8 * m, cx, ca, threads=cores)
            cc = big_circuit(1, m, Q, None)
            t0_ = time.perf_counter()
            reps_ = 0
            while True:
                orc.prove(osrs_, m, Q, cc["wL"], cc["wR"], cc["wO"], cc["cs"], cc["aL"], cc["aR"], cc["aO"], ctr, True)
                reps_ += 1
                if time.perf_counter() - t0_ > budget_s or reps_ >= max_reps:
                    break
            return (time.perf_counter() - t0_) / reps_, osrs_

        cpu_lg = args.cpu_log2n
        if cpu_lg <= 0:     # bounded sample whatever the host: probe at 2^11, then the largest n <= 2^15 whose proof stays under ~5 s
            probe, _ = cpu_prove_time(11, 0.0, 1)
            cpu_lg = 11
            while cpu_lg < 15 and probe * (1 << (cpu_lg + 1 - 11)) <= 5.0:
                cpu_lg += 1
        cn = 1 << cpu_lg
        cdt, osrs = cpu_prove_time(cpu_lg, 12.0, 4)
        cmsm_n = 1 << 16
        csc = rand_fr_array(cr, cmsm_n)
        t0 = time.perf_counter()
        orc.msm_srs(osrs, 0, -(cmsm_n // 2) if 8 * cn >= cmsm_n // 2 else -8 * cn, csc[: min(cmsm_n, 16 * cn)], 1, cores)
        cmsm_dt = time.perf_counter() - t0
        cmsm_terms = min(cmsm_n, 16 * cn)
        # the reference-shaped cost ("cpu-literal"): per-term double-and-add fold for the MSMs
        # (CommitmentScheme.hs:26-29) and the schoolbook product for tPoly, one thread as the reference never forks; small n only
        orc.set_mode(0, 1)
        lit_n = 256
        lsrs = orc.SRS(8 * lit_n, cx, ca, threads=cores)
        lc = big_circuit(1, lit_n, Q, None)
        t0 = time.perf_counter()
        orc.prove(lsrs, lit_n, Q, lc["wL"], lc["wR"], lc["wO"], lc["cs"], lc["aL"], lc["aR"], lc["aO"], ctr, False)
        lit_dt = time.perf_counter() - t0
        orc.set_mode(1, cores)
        cpu_baseline = {"value": round(1.0 / cdt, 4), "unit": "proofs/s", "cores": cores, "kind": "port",
                        "literal": {"n": lit_n, "s_per_proof": round(lit_dt, 2), "cores": 1,
                                    "note": "the oracle with the reference's algorithms (fold of per-term double-and-add, schoolbook tPoly): "
                                            "cost grows like n^2 in tPoly and 380 group operations per term in the MSMs"},
                        "sample": f"oracle/sonic_oracle.c (Pippenger + NTT, {cores} threads) prove() at n=2^{cpu_lg}, Q={Q}, d=8n; "
                                  f"{cdt:.2f}s per proof; cost is ~linear in n, so n=2^{args.log2n} would be ~{cdt * (n / cn):.0f}s per proof",
                        "msm_scalar_muls_per_s": round(cmsm_terms / cmsm_dt, 1), "msm_sample": f"N={cmsm_terms} Pippenger, {cores} threads"}

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
                               f"G1 MSM N=2^{args.msm_log2} per GPU", "n": n, "Q": Q, "d": d, "sharding": "proof-per-rank; MSM range-sharded",
                   "streaming": "K proofs streamed by one host thread through 2 prover handles per GPU (submit / collect); "
                                "the strictly sequential rate is in `sequential`"},
        "msm": {"metric": "G1 MSM scalar-muls/sec", "value": round(msm_per_s, 1), "unit": "scalar-muls/s", "N_per_gpu": msm_n,
                "ms_per_msm": round(1e3 * dt_msm / K, 3),
                "streaming": (f"K MSMs streamed over {NL} lanes per GPU (submit / collect); one at a time in `sequential`" if NL > 0 else "none (--msm-lanes 0): one MSM at a time"),
                "sequential": {"scalar_muls_per_s": round(world * msm_n * K / dt_msm_seq, 1), "ms_per_msm": round(1e3 * dt_msm_seq / K, 3),
                               "kernel_ms_per_msm": round(kern_total / K, 3), "same_result_as_streamed": seq_result == stream_result}},
        "roofline": roofline,
        "int_roofline": int_roofline,
        "cpu_baseline": cpu_baseline,
        "proof_bytes": len(proof),
        "sequential": sequential,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
