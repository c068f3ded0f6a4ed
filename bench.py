#!/usr/bin/env python3
"""bench.py -- prove() proofs/s and G1 MSM scalar-muls/s on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W     (the driver's form)
  python bench.py --gpus N ...            without RANK in the environment and N > 1: starts those N ranks itself as child processes
                                          (before this process touches the GPU), relays rank 0's JSON line and exits with their code
  python bench.py --gpus N --in-process   ONE process drives the N GPUs through the C ABI alone (sonic_prove_batch,
                                          sonic_prove_shared, sonic_msm_g1_srs_multi: no torch.distributed, no RCCL)

A "step" is one pass of the hot path over one batch of synthetic input: one full prove()
(Sonic.Protocol.prove incl. hscProve) on a synthetic random circuit (the reference's rndCircuit
generator at scale), circuit + assignment + SRS already resident in HBM when the timed region starts.
The K steps are streamed -- one host thread, two prover handles per GPU used in turn (sonic_prover_submit /
sonic_prover_collect), nothing synchronised between steps, all K proofs complete (and byte-identical to the
one-at-a-time proofs) at the closing barrier; `value` = K / that time.  Beside it, in the same line: the strictly sequential
rate (`sequential`), the same stream WITHOUT the per-circuit precomputation (`resident_unprepared`) and the reference's own call
shape -- circuit, assignment and transcript handed over as host buffers per call (`one_shot`).  Round 6: BASELINE configs[1]
(`config2`: n = 2^14, d = 2^17), configs[4] (`config5`: 64 proofs at n = 2^16 through sonic_prove_batch) and the reference's own criterion
shape (`criterion_shape`: bench/Main.hs's two examples, d = 25 n) are legs of the same line; `sensitivities.dense_weights` shows what the
headline owes to rndCircuit's all-ones rows.
Workload (BASELINE.json configs[2], "n=2^18, d=2^20"): the reference rejects d < 7n
(src/Sonic/Protocol.hs:54-55), so prove() runs at the stated n = 2^18 with d = 8n = 2^21 and the
standalone MSM runs at exactly N = d = 2^20 terms (run A); the other reading -- the stated d = 2^20 with
n = d/8 = 2^17 -- and the Q / seed sensitivities of SURVEY 8d are in `sensitivities`.  Q = 2.

N > 1: one process per GPU, SRS replicated.
  * prove() shards by proof (each rank proves its own proofs: no data-path collective)            -> `value`, weak scaling
  * `msm`: each rank one 2^20-term slice of an N*2^20-term MSM, partials all-gathered             -> weak scaling
  * `msm_strong`: ONE fixed 2^22-term MSM (BASELINE.json configs[3]) split over the N ranks: term ranges for the
    accumulation, then an all-to-all of bucket ranges so that each rank reduces 1/N of the buckets   -> strong scaling
  * `prove_strong`: ONE n = 2^20 proof shared by the N ranks (3.3-KB shares all-gathered)             -> strong scaling
  * `in_process` (rank 0, the other ranks idle on the store): the same three through the C ABI from one process
All collectives run over RCCL on device tensors (sonic_amd/distributed.py); a process group of ONE rank (launched under
torch.distributed.run with N = 1) still executes them.

Every number of the roofline objects can be recomputed from the files they name under profiles/.
Prints ONE JSON line on rank 0; exits non-zero if any leg failed (`leg_errors`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # one hardware queue per prover stream (see sonic_amd/__init__.py); before torch touches HIP

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
XGMI_LINK_GBS = 153.0          # MI355X_MICROARCH.md: 7 point-to-point links x ~153 GB/s per GPU
ROUND = "r06"


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


def scalar_muls_reference(n, Q):
    """terms of the 7 + 4Q MSMs as the reference runs them (27n + 28 + 2Q + Q(11n + Q))"""
    return 27 * n + 28 + 2 * Q + Q * (11 * n + Q)


def scalar_muls_executed(n, Q, prepared):
    """terms the kernels actually run per proof (sonic_amd/csrc/share_plan.hpp, share_line): T 7n+9, W_t 7n+8, R / W_a / W_b 3n+4 each,
    per j: S_j (n + a Q-term MSM over the committed rows when the handle is prepared, else 3n+1), W_j 3n, W'_j 3n; C 2n+Q+1 -- n + Q from
    n = 2^17, where it runs over the SRS's symmetric sums (prove.hip, sym_on) --, Q_j 2n+Q each, Q_v 2n+Q"""
    per_j = ((n + Q) if prepared else (3 * n + 1)) + 3 * n + 3 * n + (2 * n + Q)
    sym = n >= (1 << 17) and os.environ.get("SONIC_PROVE_SYM", "") != "0" and os.environ.get("SONIC_SRS_SYM", "") != "0"
    c_terms = (n + Q) if sym else (2 * n + Q + 1)
    return (7 * n + 9) + (7 * n + 8) + 3 * (3 * n + 4) + Q * per_j + c_terms + (2 * n + Q)


def xgmi_exchange_ms(bytes_per_pair, efficiency=0.7, fixed_ms=0.02):
    """MODEL of an all-to-all / all-gather step between GPUs of one node: every pair has its own xGMI link, so the step takes one
    message over one link; `efficiency` of the link rate assumed attainable + a fixed launch / synchronisation cost"""
    return 1e3 * bytes_per_pair / (efficiency * XGMI_LINK_GBS * 1e9) + fixed_ms


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=18, help="mult. gates n = 2^log2n (default: BASELINE configs[2])")
    ap.add_argument("--Q", type=int, default=2)
    ap.add_argument("--msm-log2", type=int, default=20)
    ap.add_argument("--msm-strong-log2", type=int, default=22, help="terms of the ONE MSM that is split over all ranks (BASELINE configs[3]); capped by the SRS")
    ap.add_argument("--msm-strong", action="store_true", help="only the strong-scaling MSM leg")
    ap.add_argument("--emulate-world", type=int, default=8, help="(1 GPU) also time one rank's share of the strong-scaling MSM as if there were this many "
                                                                "ranks: 1/E of the terms, a device copy in place of the all-to-all, 1/E of the buckets (0: off)")
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
    ap.add_argument("--no-north-star-cpu", action="store_true", help="skip the CPU port on the SAME n = 2^strong-log2n proof (~55 s of the run)")
    ap.add_argument("--north-star-cpu", action="store_true", help=argparse.SUPPRESS)       # (round 4's opt-in; the leg is on by default now)
    ap.add_argument("--prove-only", action="store_true", help="nothing but the proofs (PMC pass for the per-kernel instruction budget of a proof)")
    ap.add_argument("--no-sensitivities", action="store_true", help="skip the second reading (n = d/8) and the Q / seed / dense-weights sensitivities")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs[1] (n = 2^14) and configs[4] (64 proofs at n = 2^16) and the reference's criterion shape")
    ap.add_argument("--in-process", action="store_true", help="N > 1 from ONE process through the C ABI (no torch.distributed)")
    ap.add_argument("--devices", default="", help="(--in-process) comma-separated ordinals instead of 0..N-1; an ordinal may repeat (tests on one GPU)")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: the N ranks as child processes of torch.distributed.run -- started here, before
    this process has touched HIP or torch.cuda -- rank 0's JSON line relayed, the children's exit code handed on.  Never a world-1
    measurement labelled N."""
    try:
        import torch
        have = torch.cuda.device_count()           # (counting does not initialise the GPU on this image)
    except Exception:
        have = 0
    if have < args.gpus and args.backend == "nccl":
        log(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); not measuring something else under that label")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("bench.py: no RANK in the environment, starting the ranks: " + " ".join(cmd))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line:
        print(line, flush=True)
    return proc.returncode if proc.returncode else (0 if line else 4)


def kernel_times(L, K):
    names = C.create_string_buffer(16384)
    L.sonic_profile_names(names, 16384)
    out = {}
    for nm in names.value.decode().split():
        ms, cnt = C.c_double(), C.c_int64()
        L.sonic_profile_get(nm.encode(), C.byref(ms), C.byref(cnt))
        out[nm] = (ms.value, cnt.value)
    return out


class Bench:
    """What every leg needs -- the library, the process group (if any), the bench's SRS, circuit and transcripts, the helpers that keep
    ranks in step -- and one method per leg.  main() only decides which legs run and assembles the line (round 6: VERDICT r05 weak 10;
    no leg's measurement changed in the move)."""

    def __init__(self, args, argv):
        self.args = args
        launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # under torch.distributed.run: a process group even for N = 1
        if args.gpus > 1 and not launched and not args.in_process:
            sys.exit(spawn_ranks(args, argv))
        if args.in_process and launched:
            log("bench.py: --in-process drives all GPUs from ONE process; do not start it under torch.distributed.run")
            sys.exit(2)
        self.launched = launched
        self.rank = int(os.environ.get("RANK", "0")) if launched else 0
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
        local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
        if launched and self.world != args.gpus:
            log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {self.world}: refusing to label a {self.world}-rank measurement as {args.gpus}")
            sys.exit(2)

        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ndev = max(1, torch.cuda.device_count())
        self.inproc_devices = None
        if args.in_process:
            self.inproc_devices = [int(v) for v in args.devices.split(",")] if args.devices else list(range(args.gpus))
            if len(self.inproc_devices) != args.gpus or any(v < 0 or v >= self.ndev for v in self.inproc_devices):
                log(f"bench.py: --in-process --gpus {args.gpus} needs {args.gpus} device ordinals below {self.ndev} (got {self.inproc_devices})")
                sys.exit(2)
        self.dev_index = self.inproc_devices[0] if self.inproc_devices else local_rank % self.ndev
        torch.cuda.set_device(self.dev_index)
        self.device = torch.device("cuda", self.dev_index)
        self.use_nccl = args.backend == "nccl"
        if launched:
            import datetime
            # a rank that fails between collectives leaves the others waiting: minutes, not the default half hour, then the run fails loudly
            tmo = datetime.timedelta(seconds=600)
            if self.use_nccl:
                dist.init_process_group(backend="nccl", device_id=self.device, timeout=tmo)
            else:
                dist.init_process_group(backend=args.backend, timeout=tmo)
        self.pg = launched
        self.coll_dev = self.device if self.use_nccl else torch.device("cpu")

        import sonic_amd
        from sonic_amd import _lib, distributed as sd
        from sonic_amd.workload import big_circuit, rand_fr_array
        self.sonic_amd, self._lib, self.sd = sonic_amd, _lib, sd
        self.big_circuit, self.rand_fr_array = big_circuit, rand_fr_array
        self.L = _lib.lib()
        _lib.check(self.L.sonic_init(self.dev_index))

        self.n, self.Q = 1 << args.log2n, args.Q
        self.d = 8 * self.n
        self.msm_n = min(1 << args.msm_log2, 2 * self.d)          # the standalone MSM reads its points from this SRS (2d+1 per basis)
        self.strong_n = min(1 << args.msm_strong_log2, 2 * self.d)
        self.K, self.W = args.steps, args.warmup
        self.leg_errors = {}
        self.proof = b""
        self.pipe = self.circ = self.circuit = self.asg = None
        self.do_prove = not (args.msm_only or args.msm_strong)
        self.depth = 1 if (args.kernel_table or args.no_pipeline) else 2

    # ---- keeping ranks in step ----
    def barrier(self):
        if self.pg:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        self.L.sonic_device_sync()

    def max_over_ranks(self, vals):
        t = self.torch.tensor(vals, dtype=self.torch.float64, device=self.coll_dev)
        if self.pg:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def agree(self, ok: bool) -> bool:
        """every rank reports whether its side of a phase worked; all continue only if all did (a rank that raised must not
        leave the others inside the next collective).  One 4-byte all-reduce."""
        if not self.pg:
            return ok
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def run_leg(self, name, body):
        """body() -> result dict.  An exception on this rank is recorded; the ranks then agree, and a leg that failed anywhere is a
        failed leg everywhere (its collectives are over: body() either finished them or raised before / between them -- a rank stuck
        INSIDE a collective is what the process group's timeout is for)."""
        res, err = None, None
        try:
            res = body()
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        if not self.agree(err is None):
            self.leg_errors[name] = err or "failed on another rank"
            return {"error": self.leg_errors[name]}
        return res

    def rank0_leg(self, name, body):
        """a leg that only rank 0 runs (no collective inside): an exception becomes the leg's error"""
        try:
            return body()
        except Exception as e:      # noqa: BLE001
            self.leg_errors[name] = repr(e)
            return {"error": repr(e)}

    def make_transcripts(self, seed, count, q=None):
        q = self.Q if q is None else q
        rng_ = np.random.default_rng(seed)
        out = [self.rand_fr_array(rng_, 8 + 2 * q) for _ in range(count)]
        for t in out:
            t[:, 0] |= 1                                   # evaluation points must be non-zero
        return out

    def make_keys(self):
        """x and alpha of every SRS of the run (a benchmark SRS: known trapdoor, as in every test of the reference), from seed 0"""
        if hasattr(self, "x"):
            return
        seed_rng = np.random.default_rng(0)
        self.x = int.from_bytes(self.rand_fr_array(seed_rng, 1)[0].tobytes(), "little") | 1
        self.alpha = int.from_bytes(self.rand_fr_array(seed_rng, 1)[0].tobytes(), "little") | 1
        self.srs = None

    # ---- setup (untimed): SRS on the GPU, circuit resident in HBM ----
    def setup(self):
        a, S = self.args, self.sonic_amd
        t0 = time.time()
        self.make_keys()
        self.srs = S.SRS.new(self.d, self.x, self.alpha, device=self.dev_index)
        t_srs = time.time() - t0
        if self.do_prove:       # --msm-only launches nothing but the stand-alone MSMs (so that a rocprofv3 summary of it is about them)
            self.circ = self.big_circuit(1000 + self.rank, self.n, self.Q)
            c = self.circ
            self.circuit = S.ArithCircuit(S.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"])
            self.asg = S.Assignment(c["aL"], c["aR"], c["aO"])
            self.pipe = S.ProverPipeline(self.srs, self.circuit, depth=self.depth)
            self.pipe.set_assignment(self.asg)
        self.transcripts = self.make_transcripts(77 + self.rank, self.K + self.W)
        if self.rank == 0:
            log(f"setup: SRS.new(d=2^{a.log2n + 3}) {t_srs:.1f}s, circuit n=2^{a.log2n} Q={self.Q} resident")

    # ---- timed: K x prove() ----
    # The K proofs are streamed: one host thread, two prover handles used in turn (sonic_prover_submit / sonic_prover_collect), so
    # that proof i + 1 is already running while proof i is waited for and finished on the host -- nothing is synchronised between
    # steps, everything is complete at the closing barrier.  The strictly sequential number (every prove() call finished before
    # the next begins: the latency of one proof) is measured right after and reported beside it as "sequential".
    def leg_prove(self):
        a, L, K, W, pipe, transcripts = self.args, self.L, self.K, self.W, self.pipe, self.transcripts
        pipe.prove_all(transcripts[:max(W, self.depth)])           # warm-up (also grows every handle's workspaces)
        self.barrier()
        L.sonic_profile_reset()
        L.sonic_profile_enable(1 if a.kernel_table else 0)
        t0 = time.perf_counter()
        outs = pipe.prove_all(transcripts[W:W + K])
        self.barrier()
        dt = time.perf_counter() - t0
        L.sonic_profile_enable(0)
        self.proof = outs[-1] if outs else b""
        self.dt_prove = self.max_over_ranks([dt])[0]
        self.proofs_per_s = self.world * K / self.dt_prove
        if a.kernel_table and self.rank == 0:
            rows = [(ms, cnt, nm) for nm, (ms, cnt) in kernel_times(L, K).items()]
            tot = sum(r[0] for r in rows)
            for ms, cnt, nm in sorted(rows, reverse=True):
                log(f"  {nm:28s} {ms:10.2f} ms {cnt:7d} launches {100 * ms / tot:5.1f}%")
            log(f"  kernels total {tot:.1f} ms of {dt * 1e3:.1f} ms wall ({K} proofs)")
        # strictly sequential (rank 0): the same K proofs, one finished prove() call after the other, on one handle
        sequential = None
        if self.rank == 0 and K >= 1 and self.depth > 1:
            L.sonic_device_sync()
            t0 = time.perf_counter()
            for i in range(K):
                seq_proof = pipe.provers[0].prove_bytes(transcripts[W + i])
            L.sonic_device_sync()
            dts = time.perf_counter() - t0
            sequential = {"proofs_per_s_per_gpu": round(K / dts, 4), "ms_per_proof": round(1e3 * dts / K, 2),
                          "same_bytes_as_streamed": seq_proof == self.proof}
        for px in pipe.provers[1:]:
            px.close()
        return sequential

    # ---- rank 0: the same proofs without the per-circuit precomputation, and as the reference's one-shot call ----
    # `value` streams over handles that hold circuit and assignment and have committed the constraint rows once (sonic_prover_prepare).
    # The reference's  prove srs assignment circuit  (Protocol.hs:47-52) hands all of that over per call:
    #   resident_unprepared  the same stream over handles without prepared rows (every S_j a 3n-term MSM)
    #   one_shot             sonic_prove: circuit, assignment, transcript as HOST buffers per call (PCIe-inclusive; the library re-uses
    #                        the shell of the previous one-shot call: streams, workspaces, twiddle tables), one call after the other
    def leg_unprepared(self, srs=None, circuit=None, asg=None, n=None, proof=None, transcripts=None):
        S, L, K, W = self.sonic_amd, self.L, self.K, self.W
        srs, circuit, asg = srs or self.srs, circuit or self.circuit, asg or self.asg
        n = self.n if n is None else n
        proof = self.proof if proof is None else proof
        transcripts = self.transcripts if transcripts is None else transcripts
        pp = S.ProverPipeline(srs, circuit, depth=self.depth, prepare=False)
        pp.set_assignment(asg)
        pp.prove_all(transcripts[:max(W, self.depth)])
        L.sonic_device_sync()
        t0_ = time.perf_counter()
        outs_ = pp.prove_all(transcripts[W:W + K])
        L.sonic_device_sync()
        dt_ = time.perf_counter() - t0_
        pp.close()
        # a handle that is not prepared takes the runs of equal coefficients out of S_j from n = 2^16 (prove.hip, commit_runs): of rndCircuit's
        # two runs of n coefficients per S_j the whole 256-coefficient tiles -- n and n - 256 terms -- become at most four terms over the
        # SRS's running sums
        runs = n >= (1 << 16) and os.environ.get("SONIC_PROVE_RUNS", "") != "0" and os.environ.get("SONIC_SRS_PREFIX", "") != "0"
        executed = scalar_muls_executed(n, self.Q, False) - (self.Q * (2 * n - 256 - 4) if runs else 0)
        return {"proofs_per_s_per_gpu": round(K / dt_, 4), "ms_per_proof": round(1e3 * dt_ / K, 2), "same_bytes_as_prepared": outs_[-1] == proof,
                "scalar_muls_executed_per_proof": executed, "runs_of_equal_coefficients_through_running_sums": bool(runs)}

    def leg_one_shot(self, srs=None, circ=None, n=None, proof=None, transcripts=None):
        L, K, W, Q, _lib = self.L, self.K, self.W, self.Q, self._lib
        srs, circ = srs or self.srs, circ or self.circ
        n = self.n if n is None else n
        proof = self.proof if proof is None else proof
        transcripts = self.transcripts if transcripts is None else transcripts
        psz = L.sonic_proof_size(Q)
        out = C.create_string_buffer(psz)
        ptr = lambda a: a.ctypes.data       # noqa: E731

        def call(tr):
            _lib.check(L.sonic_prove(srs._h, n, Q, ptr(circ["wL"]), ptr(circ["wR"]), ptr(circ["wO"]), ptr(circ["cs"]), ptr(circ["aL"]), ptr(circ["aR"]),
                                     ptr(circ["aO"]), tr.ctypes.data, out))
            return out.raw
        t0_ = time.perf_counter()
        call(transcripts[0])
        first_ms = 1e3 * (time.perf_counter() - t0_)
        for i in range(1, max(1, W)):
            call(transcripts[i])
        L.sonic_device_sync()
        t0_ = time.perf_counter()
        for i in range(K):
            last = call(transcripts[W + i])
        dt_ = time.perf_counter() - t0_
        host_mb = (3 * Q * n + Q + 3 * n + 8 + 2 * Q) * 32 / 1e6
        return {"ms_per_proof": round(1e3 * dt_ / K, 2), "proofs_per_s_per_gpu": round(K / dt_, 4), "first_call_ms": round(first_ms, 1),
                "host_bytes_per_call_MB": round(host_mb, 1), "same_bytes_as_streamed": last == proof,
                "what": "sonic_prove(srs, n, Q, wL, wR, wO, cs, aL, aR, aO, transcript) with host buffers, one finished call after the other: "
                        "the reference's prove srs assignment circuit (Protocol.hs:47-52); PCIe upload of circuit and assignment inside every call; "
                        "first_call_ms includes making the handle (streams, workspaces, twiddle tables), which later calls re-use"}

    def leg_batch_c_abi(self):
        # the C entry point of the throughput mode (sonic_prove_batch over two handles on this GPU): the headline's stream without
        # the Python pipeline around it
        S, L, K, W = self.sonic_amd, self.L, self.K, self.W
        hs = [S.Prover(self.srs, self.circuit, prepare=True) for _ in range(2)]
        for h in hs:
            h.set_assignment(self.asg)
        S.prove_batch(hs, self.transcripts[:max(W, 2)])
        L.sonic_device_sync()
        t0_ = time.perf_counter()
        outs_ = S.prove_batch(hs, self.transcripts[W:W + K])
        dt_ = time.perf_counter() - t0_
        for h in hs:
            h.close()
        return {"proofs_per_s_per_gpu": round(K / dt_, 4), "ms_per_proof": round(1e3 * dt_ / K, 2), "handles": 2, "same_bytes_as_streamed": outs_[-1] == self.proof,
                "what": "sonic_prove_batch(provers[2], K transcripts): one host thread per handle inside the library"}

    # ---- NTT product alone on the chip (rank 0; roofline_ntt) ----
    def leg_ntt(self):
        L, _lib, n = self.L, self._lib, self.n
        na, nb = 3 * n + 5, 4 * n + 5                       # r(X,1) and r(X,y) + s(X,y): the shapes of tPoly's product (7n + 9 coefficients)
        lgM = (na + nb - 2).bit_length()
        M = 1 << lgM
        pa, pb = self.rand_fr_array(np.random.default_rng(5), na), self.rand_fr_array(np.random.default_rng(6), nb)
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
        for nm in ("k_ntt_wide_big", "k_ntt_wide", "k_ntt_local", "k_ntt_wide4", "k_ntt_local4", "k_fr_pointwise_mul"):       # (…4: the SONIC_NTT_WAVES=2 variants)
            ms, cnt = C.c_double(), C.c_int64()
            L.sonic_profile_get(nm.encode(), C.byref(ms), C.byref(cnt))
            per[nm] = {"ms_per_product": round(ms.value / reps, 4), "launches_per_product": cnt.value // reps}
        t_ms = sum(v["ms_per_product"] for v in per.values())
        pmc_ntt, pmc_ntt_src = load_profile_json("pmc_ntt.json")        # committed rocprofv3 --pmc passes of tools/ntt_time.py (per product, M = 2^21)
        passes = sum(per[k]["launches_per_product"] for k in ("k_ntt_wide_big", "k_ntt_wide", "k_ntt_local", "k_ntt_wide4", "k_ntt_local4")) // 3
        per = {k: v for k, v in per.items() if v["launches_per_product"] or k in ("k_ntt_wide_big", "k_ntt_local")}
        alg = 288.0 * M
        ntt = {"bound": "hbm", "kernel": "three radix-2 transforms of size M, the pointwise product folded into the inverse transform's first load (k_ntt_wide_big / k_ntt_wide / k_ntt_local)",
               "M": M, "achieved": round(alg / (t_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(alg / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms_per_product": round(t_ms, 4),
               "algorithmic_bytes": alg, "algorithmic_bytes_rule": "288 M: each transform reads and writes M x 32 B once (3 x 64 M) + 96 M for the pointwise product (SURVEY 8d lower bound)",
               "hbm_passes_per_transform": passes, "bytes_by_design": float((3 * passes * 64 + 96) * M),
               "kernels": per, "measured": "HIP events around every launch of sonic_poly_mul_fr_dev, alone on the chip, 5 products",
               "traffic": (pmc_ntt or {}).get("hbm_bytes_per_product") if M == (pmc_ntt or {}).get("M") else None, "traffic_source": pmc_ntt_src,
               "rocprof_ms_per_product": (pmc_ntt or {}).get("rocprof_ms_per_product"),
               "note": "the HBM roof is SURVEY 8d's framing; with the wide stages fused five or six per pass through LDS the transforms are bound by VALU issue: "
                       "~356 instructions per butterfly in the generated assembly routines sonic_ntt_bfly2_fwd / _inv (DESIGN.md section 5).  These are the kernels "
                       "prove() runs; the single-pass form of the wide stages (SONIC_NTT_BIG=1: two HBM passes per transform, 0.87 ms alone on the chip) makes "
                       "streamed proofs slower and is not the default (profiles/r05_ntt_wide_big.txt)"}
        for ptr in (da, db, do):
            L.sonic_dev_free(ptr)
        return ntt

    # ---- the scalars of the stand-alone MSM legs, resident in HBM ----
    def upload_msm_scalars(self):
        self.sc = self.rand_fr_array(np.random.default_rng(500), max(self.msm_n, self.strong_n))       # the same scalars on every rank (the strong leg splits them)
        self.dsc = C.c_void_p()
        self._lib.check(self.L.sonic_dev_alloc(32 * self.sc.shape[0], C.byref(self.dsc)))
        self._lib.check(self.L.sonic_dev_upload(self.dsc, self.sc.ctypes.data, 32 * self.sc.shape[0]))
        self.accum_ms = 0.0

    # ---- timed: standalone G1 MSM, N = 2^20 per GPU, scalars resident in HBM ----
    def leg_msm(self):
        a, L, S, sd, K, W, srs, dsc, msm_n = self.args, self.L, self.sonic_amd, self.sd, self.K, self.W, self.srs, self.dsc, self.msm_n
        rank, world = self.rank, self.world
        kern_total = 0.0
        basis, e0 = sd.msm_shard(rank, world, self.d, msm_n)
        # (1) one MSM after the other, every launch bracketed by HIP events: the dominant kernel's duration for the roofline
        #     (alone on the chip, as in the rocprofv3 summary of --msm-only) and the latency of one MSM.  The partial stays on
        #     the device, the all-gather runs over RCCL on the lane's stream, one device-to-host copy brings the gathered partials.
        one = sd.ShardedMsm(srs, rank, world, self.device)
        for _ in range(W):
            one.run_terms(basis, e0, dsc, msm_n)
        self.barrier()
        L.sonic_profile_reset()
        L.sonic_profile_enable(1)
        t0_ = time.perf_counter()
        for _ in range(K):
            seq_result = one.run_terms(basis, e0, dsc, msm_n)
        self.barrier()
        dt_seq = time.perf_counter() - t0_
        L.sonic_profile_enable(0)
        one.close()
        # (2) the same K MSMs streamed over NL lanes: MSM i + 1 is queued before MSM i is collected, so the sort and the
        #     latency-bound reduction of one run under the accumulation of the other; each MSM's partial still goes through the
        #     all-gather and the curve additions.  `msm.value` is this throughput.
        NL = max(0, a.msm_lanes)
        lanes = [S.MsmLane(self.dev_index) for _ in range(NL)]

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
                parts = sd.allgather_partials(mine, world, device=self.coll_dev if self.use_nccl else None)
                res = sd.sum_partials(parts, world)
            return res

        if NL > 0:
            msm_stream(max(W, NL))
            self.barrier()
            t0_ = time.perf_counter()
            stream_result = msm_stream(K)
            self.barrier()
            dt_ = time.perf_counter() - t0_
        else:
            stream_result, dt_ = seq_result, dt_seq
        dt_msm, dt_msm_seq = self.max_over_ranks([dt_, dt_seq])
        msm_per_s = world * msm_n * K / dt_msm
        for ln in lanes:
            ln.close()
        kt = kernel_times(L, K)
        self.accum_ms = kt.get("k_bucket_accum", (0.0, 0))[0] / max(1, kt.get("k_bucket_accum", (0.0, 1))[1])
        per_kernel = {}
        for nm, (m2, c2) in kt.items():
            kern_total += m2
            per_kernel[nm] = round(m2 / max(1, K), 4)
            if a.kernel_table and rank == 0:
                log(f"  [msm] {nm:24s} {m2 / max(1, c2):9.3f} ms/launch x{c2}")
        return {"metric": "G1 MSM scalar-muls/sec", "value": round(msm_per_s, 1), "unit": "scalar-muls/s", "N_per_gpu": msm_n, "scaling": "weak",
                "ms_per_msm": round(1e3 * dt_msm / K, 3),
                "streaming": (f"K MSMs streamed over {NL} lanes per GPU (submit / collect); one at a time in `sequential`" if NL > 0 else "none (--msm-lanes 0): one MSM at a time"),
                "sequential": {"scalar_muls_per_s": round(world * msm_n * K / dt_msm_seq, 1), "ms_per_msm": round(1e3 * dt_msm_seq / K, 3),
                               "kernel_ms_per_msm": round(kern_total / K, 3), "kernel_ms": per_kernel, "same_result_as_streamed": seq_result == stream_result}}

    # ---- timed: ONE 2^22-term MSM split over all ranks (strong scaling; BASELINE configs[3]) ----
    def leg_msm_strong(self):
        a, L, sd, K, W, srs, dsc, d, strong_n = self.args, self.L, self.sd, self.K, self.W, self.srs, self.dsc, self.d, self.strong_n
        rank, world, _lib = self.rank, self.world, self._lib
        sh = sd.ShardedMsm(srs, rank, world, self.device)
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
        self.barrier()
        t0_ = time.perf_counter()
        for _ in range(K):
            res_strong = run()
        self.barrier()
        dt_strong = self.max_over_ranks([time.perf_counter() - t0_])[0]
        out = {"metric": "one G1 MSM split over all ranks", "N_total": strong_n, "scaling": "strong", "n_gpus": world,
               "ms_per_msm": round(1e3 * dt_strong / K, 3), "value": round(strong_n * K / dt_strong, 1), "unit": "scalar-muls/s",
               "method": ("term ranges accumulated per rank, all-to-all of bucket ranges (RCCL), 1/N of the buckets reduced per rank, all-gather of the device-side results"
                          if exchange else ("term ranges, all-gather (no window tables on this SRS)" if world > 1 else
                                            "single rank: the plain MSM (baseline of the strong-scaling curve)")),
               "same_result_as_term_range_sharding": res_strong == check}
        if a.emulate_world > 1 and world == 1:
            E = a.emulate_world
            lo_e, hi_e = sd.split_range(strong_n, E, 0)

            def timed(fn):
                for _ in range(max(1, W)):
                    fn()
                L.sonic_device_sync()
                L.sonic_profile_reset()
                L.sonic_profile_enable(1)
                t0 = time.perf_counter()
                for _ in range(K):
                    fn()
                L.sonic_device_sync()
                dte_ = time.perf_counter() - t0
                L.sonic_profile_enable(0)
                return dte_, {nm: round(m2 / K, 4) for nm, (m2, _) in kernel_times(L, K).items()}
            dte, perk = timed(lambda: sh.run_buckets_emulated(0, -d, dsc, hi_e - lo_e, E))
            # the exchange the emulation leaves out is a MODELLED term and it is INSIDE the speed-up: each rank sends one slice to each of
            # the E - 1 peers, every pair on its own xGMI link, so the all-to-all takes one slice over one link (xgmi_exchange_ms)
            _, S_e = sd.exchange_layout(srs, E)
            slice_bytes = S_e * 192
            xch_ms = xgmi_exchange_ms(slice_bytes)
            share_ms = 1e3 * dte / K
            # the other way of splitting one MSM (round 6: VERDICT r05 item 2): every rank a whole MSM over its term range -- a full bucket set
            # to reduce, nothing to exchange but the 12-KB results -- timed on the same share, so that the plan can take the cheaper one
            dtt, perk_t = timed(lambda: sh.run_terms(0, -d, dsc, hi_e - lo_e))
            terms_ms = 1e3 * dtt / K + xgmi_exchange_ms(sd.DEV_PARTIAL_BYTES)
            best_ms = min(share_ms + xch_ms, terms_ms)
            out["emulated_share"] = {"world": E, "terms": hi_e - lo_e, "ms_per_share_kernels_only": round(share_ms, 3), "kernel_ms": perk,
                                     "exchange_model": {"bytes_per_pair": slice_bytes, "link_GBps": XGMI_LINK_GBS, "assumed_efficiency": 0.7, "fixed_ms": 0.02,
                                                        "ms": round(xch_ms, 3)},
                                     "ms_per_share": round(share_ms + xch_ms, 3),
                                     "speedup_vs_single": round(1e3 * (dt_strong / K) / (share_ms + xch_ms), 2),
                                     "speedup_without_the_exchange": round((dt_strong / K) / (dte / K), 2),
                                     "term_range_mode": {"ms_per_share": round(terms_ms, 3), "kernel_ms": perk_t,
                                                         "speedup_vs_single": round(1e3 * (dt_strong / K) / terms_ms, 2),
                                                         "what": "the same share as a whole MSM over its term range (full bucket set reduced on every rank, "
                                                                 "all-gather of the 12-KB device results modelled like the exchange)"},
                                     "best_mode": "bucket_ranges" if share_ms + xch_ms <= terms_ms else "term_ranges",
                                     "best_speedup_vs_single": round(1e3 * (dt_strong / K) / best_ms, 2),
                                     "note": "UNMEASURED ON MULTI-GPU HARDWARE: one GPU doing one rank's work, device copy instead of the xGMI all-to-all; "
                                             "ms_per_share and speedup_vs_single INCLUDE the modelled exchange (a model, not a measurement)"}
        sh.close()
        return out

    # ---- timed: ONE proof at the north_star size shared by all ranks (strong scaling of prove()) ----
    # Every rank holds the same circuit, assignment and transcript over its replica of the SRS, runs its cost-balanced piece of the
    # proof's 7 + 4Q MSMs (sonic_prover_set_share) and the ranks all-gather their shares (a few KB).  One rank: the plain sequential
    # prove() -- the north_star's "prove() wall-clock at n = 2^20 on 1 MI355X".
    def leg_prove_strong(self):
        a, L, S, sd, K, W, Q = self.args, self.L, self.sonic_amd, self.sd, self.K, self.W, self.Q
        rank, world, torch, dist = self.rank, self.world, self.torch, self.dist
        ns_lg = a.strong_log2n
        ns_n, ns_d = 1 << ns_lg, 8 << ns_lg
        t0_ = time.time()
        srs_ns = self.srs if ns_d == self.d else S.SRS.new(ns_d, self.x, self.alpha, device=self.dev_index)
        t_srs_ns = time.time() - t0_
        c_ns = self.circ if (ns_n == self.n and world == 1) else self.big_circuit(2000, ns_n, Q)           # the same statement on every rank
        circuit_ns = S.ArithCircuit(S.GateWeights(c_ns["wL"], c_ns["wR"], c_ns["wO"]), c_ns["cs"])
        asg_ns = S.Assignment(c_ns["aL"], c_ns["aR"], c_ns["aO"])
        sp = sd.ShardedProver(srs_ns, circuit_ns, rank, world, self.device)
        sp.set_assignment(asg_ns)
        ns_tr = self.make_transcripts(4242, K + max(1, W))
        for i in range(max(1, W)):
            sp.prove_bytes(ns_tr[i])
        self.barrier()
        t0_ = time.perf_counter()
        for i in range(K):
            ns_proof = sp.prove_bytes(ns_tr[max(1, W) + i])
        self.barrier()
        dt_ns = self.max_over_ranks([time.perf_counter() - t0_])[0]
        # what every rank spends on its own share (no collective in the timed part): the slowest and the mean over the ranks show how
        # well the plan balances on real hardware, next to the end-to-end time above
        share_stats = None
        if world > 1:
            t_sh = []
            for _ in range(3):
                L.sonic_device_sync()
                t0_ = time.perf_counter()
                sp.prove_share(ns_tr[max(1, W) + K - 1])
                t_sh.append(time.perf_counter() - t0_)
            mine_ms = 1e3 * min(t_sh)
            t = torch.tensor([mine_ms], dtype=torch.float64, device=self.coll_dev)
            tmax, tsum = t.clone(), t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            share_stats = {"slowest_share_ms": round(float(tmax.item()), 3), "mean_share_ms": round(float(tsum.item()) / world, 3)}
        same_ns = None
        if rank == 0 and world > 1:       # the same proof made by this GPU alone (untimed): the bytes must not depend on the sharing
            alone = S.Prover(srs_ns, circuit_ns, prepare=False)
            alone.set_assignment(asg_ns)
            same_ns = alone.prove_bytes(ns_tr[max(1, W) + K - 1]) == ns_proof
            alone.close()
        share_bytes = L.sonic_proof_share_size(Q)
        out = {"metric": "ONE prove() shared by all ranks", "n": ns_n, "Q": Q, "d": ns_d, "scaling": "strong", "n_gpus": world,
               "ms_per_proof": round(1e3 * dt_ns / K, 3), "value": round(K / dt_ns, 4), "unit": "proofs/s",
               "method": ("every rank builds the polynomials its pieces read and runs a contiguous, cost-balanced piece of the proof's 7+4Q MSMs "
                          "(cuts inside an MSM split its term range); one all-gather of %d-byte shares; sonic_proof_from_shares on every rank"
                          % share_bytes) if world > 1 else "single rank: the plain sequential prove() (baseline of the curve)",
               "same_bytes_as_one_gpu_alone": same_ns, "proof_bytes": len(ns_proof), "srs_new_s": round(t_srs_ns, 2), "shares": share_stats}
        if world == 1 and a.strong_emulate > 1:
            E = a.strong_emulate
            tr_e = ns_tr[max(1, W) + K - 1]
            ms_e, shares_e = [], []
            for r in range(E):
                sp.set_emulated_rank(r, E)
                sp.prove_share(tr_e)
                L.sonic_device_sync()
                t0_ = time.perf_counter()
                for _ in range(3):
                    sh_e = sp.prove_share(tr_e)
                ms_e.append(1e3 * (time.perf_counter() - t0_) / 3)
                shares_e.append(sh_e)
            t0_ = time.perf_counter()
            comb = S.proof_from_shares(Q, shares_e, tr_e)
            t_comb = 1e3 * (time.perf_counter() - t0_)
            gather_ms = xgmi_exchange_ms(share_bytes)           # the all-gather of the shares, MODELLED like the MSM's exchange and counted
            plan_e = S.share_plan(ns_n, Q, True, E)
            t_ranks = sum(1 for pieces, _ in plan_e if pieces[1][1] > pieces[1][0] or pieces[4][1] > pieces[4][0])
            out["emulated_shares"] = {"world": E, "ms_per_share": [round(v, 2) for v in ms_e], "slowest_ms": round(max(ms_e), 2),
                                      "combine_ms_host": round(t_comb, 3), "allgather_model_ms": round(gather_ms, 3),
                                      "combined_equals_whole_proof": comb == ns_proof,
                                      "speedup_vs_one_gpu": round((1e3 * dt_ns / K) / (max(ms_e) + t_comb + gather_ms), 2),
                                      "ranks_that_repeat_the_t_product": t_ranks,
                                      "note": "UNMEASURED ON MULTI-GPU HARDWARE: this one GPU ran every rank's share in turn; speedup_vs_one_gpu INCLUDES the host "
                                              "combine and a MODELLED all-gather of %d bytes per rank (one small message per xGMI link + 20 us)" % share_bytes}
        self.ns_state = dict(dt_ns=dt_ns, ns_n=ns_n, ns_d=ns_d, srs_ns=srs_ns, c_ns=c_ns, ns_tr=ns_tr, ns_proof=ns_proof, sp=sp)
        return out

    def north_star(self):
        """rank 0, one rank: the sequential n = 2^20 proof of leg_prove_strong as the north_star object, with the CPU port timed live on the same proof"""
        a, K, W, Q = self.args, self.K, self.W, self.Q
        st = self.ns_state
        dt_ns, ns_n, ns_d = st["dt_ns"], st["ns_n"], st["ns_d"]
        north_star = {"target": "prove() wall-clock at n=2^20 (d = 8n = 2^23; BASELINE states d=2^22, which Protocol.hs:54-55 rejects) on 1 MI355X, "
                                ">= 10x the CPU prove(), bit-exact", "n": ns_n, "d": ns_d, "Q": Q,
                      "ms_per_proof": round(1e3 * dt_ns / K, 3), "how": f"{K} sequential prove() calls, each finished before the next begins"}
        if not a.no_north_star_cpu and not a.no_cpu:
            # the CPU port on this very proof, timed live in this run (~55 s on the pool's 16 usable cores)
            try:
                from oracle import orc
                cores_ns = effective_cores()
                orc.set_mode(1, cores_ns)
                srs_ns, c_ns, ns_tr = st["srs_ns"], st["c_ns"], st["ns_tr"]
                o_ns = orc.SRS.from_points(ns_d, srs_ns.points(0, -ns_d, 2 * ns_d + 1), srs_ns.points(1, -ns_d, 2 * ns_d + 1))
                t0 = time.perf_counter()
                cp_ns = orc.prove(o_ns, ns_n, Q, c_ns["wL"], c_ns["wR"], c_ns["wO"], c_ns["cs"], c_ns["aL"], c_ns["aR"], c_ns["aO"], ns_tr[max(1, W) + K - 1], True)
                cdt_ns = time.perf_counter() - t0
                north_star["cpu"] = {"kind": "port", "what": "oracle/sonic_oracle.c, the repo's plain-C port (Pippenger + NTT), timed live in this run", "cores": cores_ns, "n": ns_n,
                                     "s_per_proof": round(cdt_ns, 2), "same_bytes_as_gpu_proof": cp_ns == st["ns_proof"],
                                     "gpu_over_cpu": round(cdt_ns / (dt_ns / K), 1)}
                del o_ns
            except Exception as e:      # noqa: BLE001
                self.leg_errors["north_star_cpu"] = repr(e)
                north_star["cpu"] = {"error": repr(e)}
        return north_star

    # ---- rank 0 drives ALL GPUs from this one process through the C ABI (N > 1; the other ranks idle on the store) ----
    def leg_in_process_under_pg(self):
        a, dist = self.args, self.dist
        store = dist.distributed_c10d._get_default_store()
        in_process = None
        self.barrier()
        if self.rank == 0:
            try:
                # rank i's GPU (over RCCL every rank has its own; gloo ranks may share one: the device list then repeats an ordinal)
                in_process = in_process_legs(self.sonic_amd, self.L, self._lib, self.x, self.alpha, [i % self.ndev for i in range(self.world)], a.log2n, self.Q, self.K,
                                             max(1, self.W), a.strong_log2n, self.strong_n, self.make_transcripts, self.rand_fr_array, self.big_circuit, srs0=self.srs)
            except Exception as e:      # noqa: BLE001
                self.leg_errors["in_process"] = repr(e)
                in_process = {"error": repr(e)}
            store.set("sonic_in_process_done", "1")
        else:
            import datetime
            store.wait(["sonic_in_process_done"], datetime.timedelta(seconds=1800))      # a host-side wait: no collective kernel spins on this rank's GPU meanwhile
        self.barrier()
        return in_process

    # ---- the roofline objects (rank 0) ----
    def rooflines_msm(self):
        """roofline of the dominant kernel (k_bucket_accum of the N = 2^20 MSM): algorithmic bytes = 128 B per scalar-mul (96 B affine point +
        32 B scalar) x the terms one launch covers; and its integer roof"""
        L, msm_n, accum_ms = self.L, self.msm_n, self.accum_ms
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
        L.sonic_msm_plan(self.srs._h, msm_n, C.byref(pc), C.byref(pw_), C.byref(pb))
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
        return roofline, int_roofline

    def rooflines_prove(self):
        """whole prove(): SURVEY 8d's algorithmic bytes against the time of one streamed proof; the scalar-mul rate from the terms the
        kernels actually ran (a prepared handle runs ~45n of the reference's 49n: the S_j come from the committed rows); and (round 6) the
        integer roof of the whole proof: executed bucket additions x MADs per addition against the chip's MAD issue rate"""
        n, Q, world, proofs_per_s = self.n, self.Q, self.world, self.proofs_per_s
        lgM = (7 * n + 8).bit_length()
        scalar_muls = scalar_muls_reference(n, Q)
        executed = scalar_muls_executed(n, Q, True)
        alg_p = 128.0 * scalar_muls + 288.0 * (1 << lgM)
        roofline_prove = {"bound": "hbm", "algorithmic_bytes_per_proof": alg_p, "scalar_muls_per_proof": scalar_muls, "ntt_size": 1 << lgM,
                          "rule": "128 B x (27n + 28 + 2Q + Q(11n + Q)) + 288 M (SURVEY 8d: the reference's 49n terms at Q = 2)", "achieved": round(alg_p * proofs_per_s / world / 1e9, 2),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_p * proofs_per_s / world / 1e9 / HBM_PEAK_GBS, 5),
                          "scalar_muls_executed_per_proof": executed,
                          "scalar_muls_executed_rule": "share_plan.hpp share_line: prepared handles commit S_j as an n-term + a Q-term MSM instead of 3n + 1 terms; C over the SRS's symmetric sums is n + Q terms instead of 2n + Q + 1 (from n = 2^17)",
                          "scalar_muls_per_s_inside_prove": round(executed * proofs_per_s, 1)}
        int_roofline_prove = None
        model, model_src = load_profile_json("kernel_model.json")
        if model:
            pc, pw_, pb = C.c_int(), C.c_int(), C.c_int()
            self.L.sonic_msm_plan(self.srs._h, 3 * n, C.byref(pc), C.byref(pw_), C.byref(pb))
            n_msms = 7 + 4 * Q
            # one addition per (term, window) over the window tables minus the first entry of every bucket (a copy), plus the reduction of the
            # bucket sets: ~2 full additions per bucket (14 products against the walk's 10: counted as 2.8 walk additions)
            buckets = (1 << (pc.value - 1)) * pb.value
            walk_adds = executed * pw_.value - n_msms * buckets
            reduce_adds = 2.8 * n_msms * buckets
            adds_per_s = (walk_adds + reduce_adds) * proofs_per_s / world
            int_roofline_prove = {"bound": "v_mad_u64_u32", "achieved": round(adds_per_s * model["mads_per_addition"] / 1e12, 3), "peak": model["mad_peak_per_s"] / 1e12,
                                  "unit": "TMAD/s", "frac": round(adds_per_s * model["mads_per_addition"] / model["mad_peak_per_s"], 4),
                                  "walk_additions_per_proof": walk_adds, "reduction_additions_per_proof_walk_equivalents": round(reduce_adds),
                                  "plan": {"window_bits": pc.value, "windows": pw_.value, "bucket_sets_per_msm": pb.value, "msms": n_msms},
                                  "rule": "(executed terms x windows - first entries of the buckets + 2.8 x buckets of the 7 + 4Q sets) x MADs per addition x proofs/s / peak; "
                                          "the transforms, the sort and the polynomial kernels are NOT counted (they are ~6 % of a proof's VALU instructions, profiles/r05_valu_budget.txt), "
                                          "so this is a lower bound of what the chip issued",
                                  "model": model_src}
        return roofline_prove, int_roofline_prove


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    t_main = time.time()
    args = parse_args(argv)
    b = Bench(args, argv)
    rank, world, K, W = b.rank, b.world, b.K, b.W

    # ---------------- BASELINE configs[1] and configs[4], and the reference's own criterion shape (rank 0; round 6) ----------------
    # First, on a process that holds nothing else yet: each leg makes its own SRS and handles and frees them.  (Measured after the headline's
    # legs -- a 2^21 SRS, a dozen handles made and freed, ~100 streams created on the runtime's eight hardware queues -- the same legs read
    # 10-13 % slower than alone: 3.7 against 3.3 ms at n = 2^14, 100 against 108 proofs/s at n = 2^16, profiles/r06_bench.json of the first collection.)
    config2 = config5 = criterion = None
    small_first = (rank == 0 and not (args.msm_only or args.msm_strong) and not args.prove_only and not args.kernel_table and not args.no_configs
                   and not args.in_process)
    b.make_keys()
    if small_first:
        config2 = b.rank0_leg("config2", lambda: config2_leg(b))
        config5 = b.rank0_leg("config5", lambda: config5_leg(b))
        criterion = b.rank0_leg("criterion_shape", lambda: criterion_leg(b))
        b.L.sonic_one_shot_trim(-1)
    b.barrier()
    b.setup()

    # ---------------- the whole line from ONE process (--in-process) ----------------
    if args.in_process:
        line = in_process_line(args, b.sonic_amd, b.L, b._lib, b.srs, b.x, b.alpha, b.circuit, b.asg, b.circ, b.transcripts, b.inproc_devices, b.n, b.Q, b.d, b.msm_n,
                               b.strong_n, K, W, b.make_transcripts, b.rand_fr_array, b.big_circuit)
        print(json.dumps(line), flush=True)
        sys.exit(3 if line.get("leg_errors") else 0)

    b.dt_prove, b.proofs_per_s, sequential = 1.0, 0.0, None
    if b.do_prove:
        sequential = b.leg_prove()
    b.barrier()

    resident_unprepared = one_shot = batch_c_abi = None
    small = rank == 0 and b.do_prove and not args.prove_only and not args.kernel_table
    if small:
        resident_unprepared = b.rank0_leg("resident_unprepared", b.leg_unprepared)
        one_shot = b.rank0_leg("one_shot", b.leg_one_shot)
        batch_c_abi = b.rank0_leg("batch_c_abi", b.leg_batch_c_abi)
    b.barrier()

    ntt = None
    if b.do_prove and rank == 0 and not args.prove_only:
        ntt = b.leg_ntt()

    b.upload_msm_scalars()
    msm = msm_strong = None
    if not args.msm_strong and not args.prove_only:
        msm = b.run_leg("msm", b.leg_msm)
    if not args.msm_only and not args.prove_only:
        msm_strong = b.run_leg("msm_strong", b.leg_msm_strong)

    # ---------------- stand-alone MSM with protocol-shaped scalars (rank 0) ----------------
    msm_protocol = None
    if rank == 0 and not args.msm_only and not args.prove_only and not args.msm_strong and not args.no_sensitivities:
        msm_protocol = b.rank0_leg("msm_protocol_shaped", lambda: protocol_shaped_msm(b.sonic_amd, b.L, b._lib, b.x, b.alpha, K, max(1, W)))
    b.L.sonic_dev_free(b.dsc)
    b.barrier()

    # ---------------- second reading and sensitivities (rank 0) ----------------
    sensitivities = None
    if small and not args.no_sensitivities:
        sensitivities = b.rank0_leg("sensitivities", lambda: sensitivity_legs(b))
    b.barrier()

    prove_strong = north_star = None
    b.ns_state = {}
    if b.do_prove and not args.prove_only and args.strong_log2n > 0:
        prove_strong = b.run_leg("prove_strong", b.leg_prove_strong)
        if "error" not in (prove_strong or {}) and rank == 0 and world == 1:
            north_star = b.north_star()
        if b.ns_state.get("sp") is not None:
            b.ns_state["sp"].close()
        b.ns_state.clear()

    in_process = None
    if b.pg and world > 1 and b.do_prove and not args.prove_only:
        in_process = b.leg_in_process_under_pg()

    if rank != 0:
        if b.pg:
            b.dist.destroy_process_group()
        return

    roofline = int_roofline = None
    if msm is not None and "error" not in msm:
        roofline, int_roofline = b.rooflines_msm()
    roofline_prove = int_roofline_prove = None
    if b.do_prove:
        roofline_prove, int_roofline_prove = b.rooflines_prove()

    cpu_baseline = None
    if not args.no_cpu and b.do_prove and world == 1:        # (the CPU legs run at N = 1 only: they time host cores, not GPUs)
        cpu_baseline = b.rank0_leg("cpu_baseline", lambda: cpu_baseline_leg(args, b.sonic_amd, b.srs, b.x, b.alpha, b.circ, b.transcripts, b.proof, b.sc, b.n, b.Q, b.d, b.msm_n,
                                                                             K, W, b.big_circuit, b.rand_fr_array))

    n, Q, d = b.n, b.Q, b.d
    line = {
        "metric": "prove() proofs/sec",
        "value": round(b.proofs_per_s, 4),
        "unit": "proofs/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(1e3 * b.dt_prove / K, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 limbs (Fq 12x32, Fr 8x32 Montgomery)",
        "data": "synthetic",
        "config": {"workload": f"prove(): rndCircuit n=2^{args.log2n}, Q={Q}, SRS d=2^{args.log2n + 3} (d=8n >= 7n, Protocol.hs:54); "
                               f"G1 MSM N=2^{args.msm_log2} per GPU; one G1 MSM N=2^{b.strong_n.bit_length() - 1} over all GPUs", "n": n, "Q": Q, "d": d,
                   "sharding": "proof-per-rank (value); MSM term-range-sharded (msm, weak); one MSM bucket-range-sharded (msm_strong, strong)",
                   "streaming": "K proofs streamed by one host thread through 2 prover handles per GPU (submit / collect), circuit + assignment resident, constraint rows "
                                "committed once per circuit (sonic_prover_prepare); `sequential`, `resident_unprepared` and `one_shot` are beside it",
                   "process_group": (f"{args.backend}, {world} rank(s)" if b.pg else "none (plain single-process run)")},
        "msm": msm,
        "msm_strong": msm_strong,
        "msm_protocol_shaped": msm_protocol,
        "roofline": roofline,
        "int_roofline": int_roofline,
        "roofline_ntt": ntt,
        "roofline_prove": roofline_prove,
        "int_roofline_prove": int_roofline_prove,
        "cpu_baseline": cpu_baseline,
        "prove_strong": prove_strong,
        "north_star": north_star,
        "proof_bytes": len(b.proof),
        "sequential": sequential,
        "resident_unprepared": resident_unprepared,
        "one_shot": one_shot,
        "batch_c_abi": batch_c_abi,
        "sensitivities": sensitivities,
        "config2": config2,
        "config5": config5,
        "criterion_shape": criterion,
        "in_process": in_process,
        "status": "failed legs: " + ", ".join(sorted(b.leg_errors)) if b.leg_errors else "ok",
        "leg_errors": b.leg_errors or None,
        "wall_s": round(time.time() - t_main, 1),       # the whole run, SRS generation and the CPU legs included
    }
    print(json.dumps(line), flush=True)
    if b.pg:
        b.dist.destroy_process_group()
    if b.leg_errors:
        sys.exit(3)


# ======================================================================================================================================
def protocol_shaped_msm(sonic_amd, L, _lib, x, alpha, K, W, which=("W_t_quotient", "s_of_X_y_coefficients"), uniform=True):
    """the stand-alone MSM again with the scalars the protocol really feeds it -- the quotient w = (t(X,y) - t(z,y)) / (X - z)
    of W_t = openPoly(t(X,y), z) (Protocol.hs:81) at n = 2^17: 7n + 8 = 917 512 terms over the stated d = 2^20 SRS -- and with the
    coefficients of s(X,y) (Signature.hs:42, unprepared S_j: 3n + 1 terms of which 2n are copies of two values and n are zero-free runs),
    the shape that produces heavy buckets.  The polynomials are built on the host with python integers and the product's own NTT
    (sonic_amd.workload.wt_quotient_scalars); nothing of the oracle is involved."""
    from sonic_amd.workload import wt_quotient_scalars
    lg = 17
    n, d = 1 << lg, 8 << lg
    t0 = time.time()
    srs = sonic_amd.SRS.new(d, x, alpha)
    wt, sy = wt_quotient_scalars(sonic_amd, 3000, n, 2)
    t_setup = time.time() - t0
    out = {"n": n, "d": d, "setup_s": round(t_setup, 2),
           "what": "sonic_msm_g1_srs_dev over the plain basis from exponent -4n-8 (W_t's slice) / the alpha basis (S_j's slice), scalars resident in HBM, one MSM at a time"}
    # W_t: plain basis, exponents [-4n-8, 3n-1] (CommitmentScheme.hs:45-47).  S_j = commitPoly(d, s(X,y_j)): alpha basis, shift d - max = 0,
    # exponents [-n, 2n]; the coefficient at exponent 0 is zero and meets the omitted g^alpha (SRS.hs:38)
    for name, scal, basis, e0 in (("W_t_quotient", wt, 0, -4 * n - 8), ("s_of_X_y_coefficients", sy, 1, -n)):
        if name not in which:
            continue
        N = scal.shape[0]
        dptr = C.c_void_p()
        _lib.check(L.sonic_dev_alloc(32 * N, C.byref(dptr)))
        _lib.check(L.sonic_dev_upload(dptr, scal.ctypes.data, 32 * N))
        res = C.create_string_buffer(96)
        for _ in range(W):
            _lib.check(L.sonic_msm_g1_srs_dev(srs._h, basis, e0, dptr, N, res))
        L.sonic_device_sync()
        t0 = time.perf_counter()
        for _ in range(K):
            _lib.check(L.sonic_msm_g1_srs_dev(srs._h, basis, e0, dptr, N, res))
        dt = time.perf_counter() - t0
        uniq = int(np.unique(np.ascontiguousarray(scal).view(np.dtype((np.void, 32)))).shape[0])
        zeros = int((~scal.any(axis=1)).sum())
        # the same number of uniform scalars over the same slice, for comparison
        dtu = None
        if uniform:
            uni = np.random.default_rng(9).integers(0, 256, size=(N, 32), dtype=np.uint8)
            uni[:, 31] &= 0x3f
            _lib.check(L.sonic_dev_upload(dptr, uni.ctypes.data, 32 * N))
            for _ in range(W):
                _lib.check(L.sonic_msm_g1_srs_dev(srs._h, basis, e0, dptr, N, res))
            L.sonic_device_sync()
            t0 = time.perf_counter()
            for _ in range(K):
                _lib.check(L.sonic_msm_g1_srs_dev(srs._h, basis, e0, dptr, N, res))
            dtu = time.perf_counter() - t0
        L.sonic_dev_free(dptr)
        out[name] = {"N": N, "ms_per_msm": round(1e3 * dt / K, 3), "scalar_muls_per_s": round(N * K / dt, 1), "distinct_scalars": uniq, "zero_scalars": zeros,
                     "uniform_scalars_same_slice_ms": round(1e3 * dtu / K, 3) if dtu is not None else None}
    srs.close()
    return out


def dense_circuit(rand_fr_array, seed, n, Q):
    """a satisfied circuit whose weights are ALL uniformly random (3 Q n field elements): nothing repeats, so neither the runs of equal
    coefficients (unprepared handles) nor a cheap sum help -- the shape a general-purpose constraint system hands over.  aO = aL o aR,
    cs = wL aL + wR aR + wO aO (test/Test/Reference.hs:138,164-169), computed with python integers."""
    from sonic_amd.workload import R, fr_bytes
    rng = np.random.default_rng(seed)
    aL, aR = rand_fr_array(rng, n), rand_fr_array(rng, n)
    la = [int.from_bytes(aL[i].tobytes(), "little") for i in range(n)]
    lb = [int.from_bytes(aR[i].tobytes(), "little") for i in range(n)]
    lo = [a * b % R for a, b in zip(la, lb)]
    W = [rand_fr_array(rng, Q * n) for _ in range(3)]
    cs = []
    for q in range(Q):
        acc = 0
        for w, v in zip(W, (la, lb, lo)):
            row = w[q * n:(q + 1) * n]
            acc += sum(int.from_bytes(row[i].tobytes(), "little") * v[i] for i in range(n))
        cs.append(acc % R)
    return dict(wL=W[0], wR=W[1], wO=W[2], cs=fr_bytes(cs), aL=aL, aR=aR, aO=fr_bytes(lo), rows=None)


def sensitivity_legs(b):
    """prove() at the STATED d with n_eff = d/8 (configs[2] read the other way: n = 2^17, d = 2^20), Q in {1, 4} and seeds 1, 2 at
    the headline size; each streamed over two prepared handles like `value`.  Round 6, `dense_weights`: the headline size with uniformly
    random weights -- streamed over prepared handles, streamed over handles that are not prepared, and as the one-shot call."""
    sonic_amd, L, srs, x, alpha, n, Q, d, K, W = b.sonic_amd, b.L, b.srs, b.x, b.alpha, b.n, b.Q, b.d, b.K, max(1, b.W)
    make_transcripts, big_circuit = b.make_transcripts, b.big_circuit

    def stream(srs_, n_, q_, seed):
        c = big_circuit(seed, n_, q_)
        circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"])
        pp = sonic_amd.ProverPipeline(srs_, circuit, depth=2)
        pp.set_assignment(sonic_amd.Assignment(c["aL"], c["aR"], c["aO"]))
        trs = make_transcripts(900 + seed, K + max(W, 2), q_)
        pp.prove_all(trs[:max(W, 2)])
        L.sonic_device_sync()
        t0 = time.perf_counter()
        pp.prove_all(trs[max(W, 2):])
        L.sonic_device_sync()
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in range(K):
            pp.provers[0].prove_bytes(trs[max(W, 2) + i])
        dts = time.perf_counter() - t0
        pp.close()
        return {"n": n_, "Q": q_, "d": int(srs_.srsD), "seed": seed, "proofs_per_s": round(K / dt, 4), "ms_per_proof": round(1e3 * dt / K, 2),
                "sequential_ms_per_proof": round(1e3 * dts / K, 2), "rows_with_ones": c["rows"]}
    out = {}
    n2, d2 = n // 2, d // 2
    srs2 = sonic_amd.SRS.new(d2, x, alpha)
    out["stated_d_reading"] = dict(stream(srs2, n2, Q, 1000), note=f"BASELINE configs[2] read as d = 2^{d2.bit_length() - 1} with n = d/8 (SURVEY 8d (ii))")
    srs2.close()
    out["Q1"] = stream(srs, n, 1, 1000)
    out["Q4"] = stream(srs, n, 4, 1000)
    out["seed1"] = stream(srs, n, Q, 1)
    out["seed2"] = stream(srs, n, Q, 2)
    # dense weights: what the three ways of proving cost when no row of the circuit repeats a value
    t0 = time.time()
    dc = dense_circuit(b.rand_fr_array, 31, n, Q)
    t_gen = time.time() - t0
    circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(dc["wL"], dc["wR"], dc["wO"]), dc["cs"])
    asg = sonic_amd.Assignment(dc["aL"], dc["aR"], dc["aO"])
    trs = make_transcripts(931, K + max(W, 2))
    pp = sonic_amd.ProverPipeline(srs, circuit, depth=2)
    pp.set_assignment(asg)
    pp.prove_all(trs[:max(W, 2)])
    L.sonic_device_sync()
    t0 = time.perf_counter()
    outs = pp.prove_all(trs[max(W, 2):])
    L.sonic_device_sync()
    dt = time.perf_counter() - t0
    pp.close()
    keepW, keepT = b.W, b.transcripts
    try:
        b.W, b.transcripts = max(W, 2), trs          # (the shared legs read their warm-up count and transcripts from the context)
        unprep = b.leg_unprepared(circuit=circuit, asg=asg, proof=outs[-1], transcripts=trs)
        shot = b.leg_one_shot(circ=dc, proof=outs[-1], transcripts=trs)
    finally:
        b.W, b.transcripts = keepW, keepT
    out["dense_weights"] = {"n": n, "Q": Q, "weights": "3 Q n uniformly random field elements (no repeated value in any row)", "circuit_generation_s": round(t_gen, 2),
                            "prepared_streamed": {"proofs_per_s": round(K / dt, 4), "ms_per_proof": round(1e3 * dt / K, 2)},
                            "resident_unprepared": {k: unprep[k] for k in ("proofs_per_s_per_gpu", "ms_per_proof", "same_bytes_as_prepared")},
                            "one_shot": {k: shot[k] for k in ("proofs_per_s_per_gpu", "ms_per_proof", "host_bytes_per_call_MB", "same_bytes_as_streamed")},
                            "note": "the default rows are all ones (test/Test/Reference.hs:141-155): an unprepared handle commits their runs through running sums of "
                                    "the SRS (2 terms per run) and sum_q y^{n+q} w_q is cheap; with dense weights neither applies -- the prepared stream is unaffected "
                                    "(its S_j come from the committed rows either way)"}
    return out


def _stream_and_sequential(b, srs_, circuit, asg, trs, warm, timed_count):
    """K proofs streamed over two prepared handles, then the same proofs one finished call after the other on the first handle"""
    S, L = b.sonic_amd, b.L
    pp = S.ProverPipeline(srs_, circuit, depth=2)
    pp.set_assignment(asg)
    pp.prove_all(trs[:warm])
    L.sonic_device_sync()
    t0 = time.perf_counter()
    outs = pp.prove_all(trs[warm:warm + timed_count])
    L.sonic_device_sync()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    seq = [pp.provers[0].prove_bytes(t) for t in trs[warm:warm + timed_count]]
    dts = time.perf_counter() - t0
    pp.close()
    return dt, dts, outs, seq


def config2_leg(b):
    """BASELINE configs[1]: "n = 2^14 random circuit, SRS d = 2^16" -- the reference rejects d < 7n (Protocol.hs:54-55), so d = 8n = 2^17
    (SURVEY 8d).  Streamed over two prepared handles and one finished call after the other; 40 proofs each after 6 warm-ups."""
    S = b.sonic_amd
    lg, Q = 14, b.Q
    n, d = 1 << lg, 8 << lg
    srs2 = b.srs if (b.srs is not None and d == b.d) else S.SRS.new(d, b.x, b.alpha, device=b.dev_index)
    c = b.big_circuit(1400, n, Q)
    circuit = S.ArithCircuit(S.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"])
    asg = S.Assignment(c["aL"], c["aR"], c["aO"])
    cnt, warm = 40, 6
    trs = b.make_transcripts(1401, cnt + warm)
    dt, dts, outs, seq = _stream_and_sequential(b, srs2, circuit, asg, trs, warm, cnt)
    pc, pw_, pb = C.c_int(), C.c_int(), C.c_int()
    b.L.sonic_msm_plan(srs2._h, 3 * n, C.byref(pc), C.byref(pw_), C.byref(pb))
    if srs2 is not b.srs:
        srs2.close()
    executed = scalar_muls_executed(n, Q, True)
    return {"workload": f"prove(): rndCircuit n=2^{lg}, Q={Q}, SRS d=2^{lg + 3} (BASELINE configs[1] states d=2^16 = 4n, which Protocol.hs:54-55 rejects)", "n": n, "Q": Q, "d": d,
            "streamed": {"ms_per_proof": round(1e3 * dt / cnt, 3), "proofs_per_s": round(cnt / dt, 2)},
            "sequential": {"ms_per_proof": round(1e3 * dts / cnt, 3), "proofs_per_s": round(cnt / dts, 2)},
            "proofs_timed": cnt, "same_bytes_streamed_and_sequential": outs == seq,
            "plan": {"window_bits": pc.value, "windows": pw_.value, "bucket_sets_per_msm": pb.value,
                     "one_chain_per_proof": "the 7 + 4Q MSMs of a proof run as ONE batched kernel chain (prove.hip, fused)"},
            "scalar_muls_executed_per_proof": executed, "scalar_muls_per_s_inside_prove_streamed": round(executed * cnt / dt, 1)}


def config5_leg(b):
    """BASELINE configs[4]: "batch of 64 independent proofs at n = 2^16, d = 2^18 streamed" -- d = 8n = 2^19 (Protocol.hs:54-55) -- through the C
    entry point of the throughput mode, sonic_prove_batch, over two prepared handles on this GPU, every proof its own assignment and
    transcript handed over as host buffers (64 x 6.3 MB).  The reference's `mapM (prove srs) assignments` for one circuit
    (Protocol.hs:47-52).  A sample of the batch is compared with one handle proving alone."""
    S, L = b.sonic_amd, b.L
    lg, Q, cnt = 16, b.Q, 64
    n, d = 1 << lg, 8 << lg
    srs5 = b.srs if (b.srs is not None and d == b.d) else S.SRS.new(d, b.x, b.alpha, device=b.dev_index)
    base = b.big_circuit(1600, n, Q)
    circuit = S.ArithCircuit(S.GateWeights(base["wL"], base["wR"], base["wO"]), base["cs"])
    # (rndCircuit's constants are cs = w . a: another assignment would need another circuit, and sonic_prove_batch proves ONE circuit; the batch
    # uploads its assignment per proof -- the bytes cross PCIe 64 times as 64 different ones would -- and every proof has its own transcript)
    asg = S.Assignment(base["aL"], base["aR"], base["aO"])
    hs = [S.Prover(srs5, circuit, prepare=True) for _ in range(2)]
    trs = b.make_transcripts(1601, cnt + 4)
    S.prove_batch(hs, trs[:4], assignments=[asg] * 4)
    L.sonic_device_sync()
    t0 = time.perf_counter()
    outs = S.prove_batch(hs, trs[4:], assignments=[asg] * cnt)
    dt = time.perf_counter() - t0
    # resident assignment (no per-proof upload): what the headline's stream does
    for h in hs:
        h.set_assignment(asg)
    S.prove_batch(hs, trs[:4])
    L.sonic_device_sync()
    t0 = time.perf_counter()
    outs_res = S.prove_batch(hs, trs[4:])
    dt_res = time.perf_counter() - t0
    alone = [hs[0].prove_bytes(trs[4 + i]) for i in (0, 1, cnt - 1)]
    same = [outs[0], outs[1], outs[cnt - 1]] == alone and outs_res == outs
    for h in hs:
        h.close()
    if srs5 is not b.srs:
        srs5.close()
    executed = scalar_muls_executed(n, Q, True)
    return {"workload": f"{cnt} proofs of ONE rndCircuit n=2^{lg}, Q={Q}, SRS d=2^{lg + 3}, through sonic_prove_batch over 2 prepared handles on this GPU (BASELINE configs[4]; d = 8n)",
            "n": n, "Q": Q, "d": d, "proofs": cnt,
            "assignment_per_proof_from_host": {"proofs_per_s_per_gpu": round(cnt / dt, 2), "ms_per_proof": round(1e3 * dt / cnt, 3), "host_MB_per_proof": round(3 * n * 32 / 1e6, 1)},
            "assignment_resident": {"proofs_per_s_per_gpu": round(cnt / dt_res, 2), "ms_per_proof": round(1e3 * dt_res / cnt, 3)},
            "same_bytes_as_one_handle_alone": same,
            "scalar_muls_executed_per_proof": executed, "scalar_muls_per_s_inside_prove": round(executed * cnt / dt_res, 1),
            "note": "proofs/s of ONE GPU; over N GPUs the batch is split by proof with no collective (SURVEY 8e), two handles per device"}


def criterion_leg(b):
    """the reference's own benchmark (bench/Main.hs:18-50): arithCircuitExample1 / 2 -- one and two multiplication gates --, x = 1, alpha = 4,
    d = 25 n; `prove` alone per call (sonic_amd.prove: the one-shot entry point with host buffers, like the reference's call) and the
    reference's timed closure, SRS.new + prove.  These are plumbing sizes: they show the fixed cost of a call.  The circuits are
    bench/Main.hs's constants restated in examples/main.py (n = 2) and its one-gate sibling below; the proofs are verified."""
    S = b.sonic_amd
    R = S.R_MODULUS
    import importlib.util
    spec = importlib.util.spec_from_file_location("sonic_example_main", os.path.join(ROOT, "examples", "main.py"))
    example = importlib.util.module_from_spec(spec)          # examples/main.py: arithCircuitExample of examples/Main.hs:38-63
    spec.loader.exec_module(example)
    out = {"what": "ms per call, mean of 30 calls after 5 warm-ups; d = 25 n, x = 1, alpha = 4 (bench/Main.hs:18-27)"}
    # arithCircuitExample1 (test/Test/Reference.hs:38-50): ONE gate, two linear constraints aL = 10, aR = 12
    ex1 = (S.ArithCircuit(S.GateWeights([[1], [0]], [[0], [1]], [[0], [0]]), [10, 12]), S.Assignment([10], [12], [120]))
    ex2 = example.arith_circuit_example(12)
    for name, (circuit, asg) in (("example_n1_Q2", ex1), ("example_n2_Q5", ex2)):
        n, Q = len(asg.aL), len(circuit.cs)
        d = 25 * n
        tr = [int.from_bytes(t.tobytes(), "little") % R or 1 for t in b.make_transcripts(50 + n, 1, Q)[0]]
        srs = S.SRS.new(d, 1, 4, device=b.dev_index)
        for _ in range(5):
            proof, ro = S.prove(srs, asg, circuit, transcript=tr)
        t0 = time.perf_counter()
        for _ in range(30):
            proof, ro = S.prove(srs, asg, circuit, transcript=tr)
        t_prove = 1e3 * (time.perf_counter() - t0) / 30
        ok = S.verify(srs, circuit, proof, ro.rndOracleY, ro.rndOracleZ, ro.rndOracleYZs)
        srs.close()
        t0 = time.perf_counter()
        for _ in range(5):
            s2 = S.SRS.new(d, 1, 4, device=b.dev_index)
            S.prove(s2, asg, circuit, transcript=tr)
            s2.close()
        t_both = 1e3 * (time.perf_counter() - t0) / 5
        out[name] = {"n": n, "Q": Q, "d": d, "prove_ms": round(t_prove, 3), "srs_new_plus_prove_ms": round(t_both, 2), "verified": bool(ok)}
    return out




def cpu_baseline_leg(args, sonic_amd, srs, x, alpha, circ, transcripts, proof, sc, n, Q, d, msm_n, K, W, big_circuit, rand_fr_array):
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
    return {"value": round(1.0 / cdt, 5), "unit": "proofs/s", "cores": cores, "kind": "port",
            "n": 1 << cpu_lg, "at_bench_size": cpu_lg == args.log2n, "s_per_proof": round(cdt, 2), "same_bytes_as_gpu_proof": same,
            "sample": f"oracle/sonic_oracle.c (Pippenger + NTT, {cores} threads = usable host cores of {os.cpu_count()} visible) ONE prove() at n=2^{cpu_lg}, Q={Q}, d=8n: "
                      f"{cdt:.2f}s; {srs_note}; sized by a {probe:.2f}s probe at n=2^{plg} against a {args.cpu_budget_s:.0f}s budget",
            "msm_scalar_muls_per_s": round(cmsm_n / cmsm_dt, 1), "msm_sample": f"ONE N={cmsm_n} Pippenger MSM, {cores} threads, {cmsm_dt:.2f}s",
            "literal": {"n": lit_n, "s_per_proof": round(lit_dt, 2), "cores": 1,
                        "note": "the oracle with the reference's algorithms (fold of per-term double-and-add, schoolbook tPoly): "
                                "cost grows like n^2 in tPoly and 380 group operations per term in the MSMs"}}


def in_process_legs(sonic_amd, L, _lib, x, alpha, devices, log2n, Q, K, W, strong_log2n, strong_n, make_transcripts, rand_fr_array, big_circuit, srs0=None):
    """ONE host process, every GPU in `devices`, nothing but the C ABI: replicas by SRS.new on each device (concurrently) or copied
    device to device, then
      proofs        sonic_prove_batch over two prepared handles per device (weak scaling: K proofs per device)
      prove_strong  sonic_prove_shared: one n = 2^strong_log2n proof over one handle per device
      msm_strong    sonic_msm_g1_srs_multi_dev by bucket range: one strong_n-term MSM, slices resident on the devices
    Bytes are compared with one device alone."""
    import threading
    world = len(devices)
    n, d = 1 << log2n, 8 << log2n
    out = {"devices": devices, "what": "one host process drives every GPU through include/sonic_hip.h alone (no torch.distributed, no RCCL): one host thread per handle inside the library"}
    t0 = time.time()
    reps = [None] * world
    errs = []

    def mk(i):
        try:
            if i == 0 and srs0 is not None and srs0.device == devices[0]:
                reps[i] = srs0
            else:
                reps[i] = sonic_amd.SRS.new(d, x, alpha, device=devices[i])
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=mk, args=(i,)) for i in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise RuntimeError("SRS.new on the devices failed: " + "; ".join(errs))
    out["srs_new_s_all_devices_concurrently"] = round(time.time() - t0, 2)
    # -- throughput: K proofs per device over 2 handles per device
    c = big_circuit(1000, n, Q)
    circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"])
    asg = sonic_amd.Assignment(c["aL"], c["aR"], c["aO"])
    handles = []
    for r in reps:
        for _ in range(2):
            h = sonic_amd.Prover(r, circuit, prepare=True)
            h.set_assignment(asg)
            handles.append(h)
    # proof i runs on handle i % len(handles): interleave the devices so that consecutive proofs go to different GPUs
    handles = handles[0::2] + handles[1::2]
    trs = make_transcripts(77, (K + max(W, 2)) * world)
    sonic_amd.prove_batch(handles, trs[:max(W, 2) * world])
    t0 = time.perf_counter()
    proofs = sonic_amd.prove_batch(handles, trs[max(W, 2) * world:])
    dt = time.perf_counter() - t0
    alone = handles[0].prove_bytes(trs[-1])
    out["proofs"] = {"metric": "prove() proofs/sec", "value": round(K * world / dt, 4), "ms_per_proof_per_gpu": round(1e3 * dt / K, 2), "n": n, "Q": Q, "scaling": "weak",
                     "proofs_timed": K * world, "handles": len(handles), "same_bytes_as_one_handle_alone": proofs[-1] == alone,
                     "entry": "sonic_prove_batch"}
    for h in handles:
        h.close()
    # -- one MSM of strong_n terms by bucket range
    sc = rand_fr_array(np.random.default_rng(500), strong_n)
    slices = []
    for i, r in enumerate(reps):
        lo, hi = strong_n * i // world, strong_n * (i + 1) // world
        ptr = C.c_void_p()
        _lib.check(L.sonic_dev_alloc_on(devices[i], 32 * max(1, hi - lo), C.byref(ptr)))
        if hi > lo:
            _lib.check(L.sonic_dev_upload(ptr, sc[lo:hi].ctypes.data, 32 * (hi - lo)))
        slices.append((-d + lo, ptr, hi - lo))
    res = {}
    for mode, name in ((1, "bucket_ranges"), (0, "term_ranges")):
        try:
            for _ in range(W):
                got = sonic_amd.msm_g1_srs_multi(reps, 0, -d, None, mode=mode, d_slices=slices)
            t0 = time.perf_counter()
            for _ in range(K):
                got = sonic_amd.msm_g1_srs_multi(reps, 0, -d, None, mode=mode, d_slices=slices)
            dtm = time.perf_counter() - t0
            res[name] = {"ms_per_msm": round(1e3 * dtm / K, 3), "scalar_muls_per_s": round(strong_n * K / dtm, 1), "result": got.hex()[:16]}
        except Exception as e:      # noqa: BLE001
            res[name] = {"error": repr(e)}
    one_ptr = C.c_void_p()
    _lib.check(L.sonic_dev_alloc_on(devices[0], 32 * strong_n, C.byref(one_ptr)))
    _lib.check(L.sonic_dev_upload(one_ptr, sc.ctypes.data, 32 * strong_n))
    one = C.create_string_buffer(96)
    for _ in range(W):
        _lib.check(L.sonic_msm_g1_srs_dev(reps[0]._h, 0, -d, one_ptr, strong_n, one))
    t0 = time.perf_counter()
    for _ in range(K):
        _lib.check(L.sonic_msm_g1_srs_dev(reps[0]._h, 0, -d, one_ptr, strong_n, one))
    dt1 = time.perf_counter() - t0
    L.sonic_dev_free(one_ptr)
    for _, ptr, _ in slices:
        L.sonic_dev_free(ptr)
    for v in res.values():
        if "result" in v:
            v["same_result_as_one_gpu"] = v.pop("result") == one.raw.hex()[:16]
            v["speedup_vs_one_gpu"] = round((1e3 * dt1 / K) / v["ms_per_msm"], 2)
    out["msm_strong"] = dict(res, N_total=strong_n, one_gpu_ms_per_msm=round(1e3 * dt1 / K, 3), scaling="strong", entry="sonic_msm_g1_srs_multi_dev")
    for i, r in enumerate(reps):
        if r is not srs0:
            r.close()
    # -- one proof at the north-star size over one handle per device
    if strong_log2n > 0:
        ns_n, ns_d = 1 << strong_log2n, 8 << strong_log2n
        reps2 = [None] * world

        def mk2(i):
            try:
                reps2[i] = sonic_amd.SRS.new(ns_d, x, alpha, device=devices[i])
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))
        th = [threading.Thread(target=mk2, args=(i,)) for i in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise RuntimeError("SRS.new (north-star size) failed: " + "; ".join(errs))
        c2 = big_circuit(2000, ns_n, Q)
        circuit2 = sonic_amd.ArithCircuit(sonic_amd.GateWeights(c2["wL"], c2["wR"], c2["wO"]), c2["cs"])
        asg2 = sonic_amd.Assignment(c2["aL"], c2["aR"], c2["aO"])
        hs = []
        for r in reps2:
            h = sonic_amd.Prover(r, circuit2, prepare=True)
            h.set_assignment(asg2)
            hs.append(h)
        tr2 = make_transcripts(4242, K + W)
        for i in range(W):
            sonic_amd.prove_shared(hs, tr2[i])
        t0 = time.perf_counter()
        for i in range(K):
            shared = sonic_amd.prove_shared(hs, tr2[W + i])
        dts = time.perf_counter() - t0
        hs[0].set_share(0, 1)
        hs[0].prove_bytes(tr2[W + K - 1])
        t0 = time.perf_counter()
        for i in range(K):
            alone2 = hs[0].prove_bytes(tr2[W + i])
        dta = time.perf_counter() - t0
        out["prove_strong"] = {"n": ns_n, "d": ns_d, "Q": Q, "scaling": "strong", "ms_per_proof": round(1e3 * dts / K, 3), "one_gpu_ms_per_proof": round(1e3 * dta / K, 3),
                               "speedup_vs_one_gpu": round(dta / dts, 2), "same_bytes_as_one_gpu_alone": shared == alone2, "entry": "sonic_prove_shared"}
        for h in hs:
            h.close()
        for r in reps2:
            r.close()
    return out


def in_process_line(args, sonic_amd, L, _lib, srs, x, alpha, circuit, asg, circ, transcripts, devices, n, Q, d, msm_n, strong_n, K, W, make_transcripts, rand_fr_array,
                    big_circuit):
    """`bench.py --gpus N --in-process`: the contract's line with every leg made by ONE process through the C ABI"""
    errs = {}
    try:
        legs = in_process_legs(sonic_amd, L, _lib, x, alpha, devices, args.log2n, Q, K, max(1, W), args.strong_log2n, strong_n, make_transcripts, rand_fr_array, big_circuit, srs0=srs)
    except Exception as e:      # noqa: BLE001
        legs = {"error": repr(e)}
        errs["in_process"] = repr(e)
    pr = legs.get("proofs", {})
    world = len(devices)
    return {"metric": "prove() proofs/sec", "value": pr.get("value", 0.0), "unit": "proofs/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": pr.get("ms_per_proof_per_gpu"), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 limbs (Fq 12x32, Fr 8x32 Montgomery)", "data": "synthetic",
            "config": {"workload": f"prove(): rndCircuit n=2^{args.log2n}, Q={Q}, SRS d=2^{args.log2n + 3} (d=8n >= 7n, Protocol.hs:54); one G1 MSM N=2^{strong_n.bit_length() - 1} over all GPUs; "
                                   f"one prove() n=2^{args.strong_log2n} over all GPUs", "n": n, "Q": Q, "d": d,
                       "sharding": "proof-per-device (value: sonic_prove_batch); one MSM bucket-range-sharded (sonic_msm_g1_srs_multi_dev); one proof shared (sonic_prove_shared)",
                       "process_group": f"none: ONE process, devices {devices}, host threads inside libsonic_hip.so"},
            "in_process": legs, "status": "ok" if not errs else "failed legs: in_process", "leg_errors": errs or None}


if __name__ == "__main__":
    main()
