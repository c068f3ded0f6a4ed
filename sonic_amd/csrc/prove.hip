// Host orchestration of the Sonic prover on one MI355X: Sonic.Protocol.prove
// (src/Sonic/Protocol.hs:47-109) with Sonic.Signature.hscProve (src/Sonic/Signature.hs:38-72),
// plus the C-ABI faces of Sonic.SRS.new, commitPoly and openPoly.
//
// Everything between "inputs in HBM" and "7+4Q window-sum slots + 3+2Q field elements in HBM" is
// device work queued with no host synchronisation: the `rnd` draws are explicit (transcript), so no
// step waits for a result.  The main stream builds the polynomials, a side stream the t(X,y) product,
// and the MSMs that depend on one polynomial run as one batched chain on an MSM lane.
// The host then runs the O(W) tails of the MSMs (msm_finish_host) and lays out the proof bytes.
//
// Each reference step is done once: the reference re-evaluates evalY 1 polyR' three times
// (Protocol.hs:63,79,80), evalY y tXY twice (:72,81) and evalY y_j sXY twice per j
// (Signature.hs:41,54).
#include "prover.hpp"

namespace sonic {

static int prove_segment(const MsmPlan& pl, int k, bool last) {
  const bool big = pl.NB >= (1 << 18);
  if (last) return big ? 16 : 4;
  return k > 1 ? (big ? 64 : 16) : (big ? 8 : 4);
}

// commitPoly (CommitmentScheme.hs:20-33): F = sum_e c_e * A[e + d - max], A = alpha basis.
// `poly` dense over exponents [lo, lo+len).  Terms whose shifted exponent leaves [-d, d] or hits the
// e' = 0 hole are legal only if their coefficient is zero (the reference's normalised sparse form
// would not contain them); otherwise FLAG_SRS_INDEX (`index` panics, CommitmentScheme.hs:70-73).
// Queues the checks and returns the MSM that is left to run.
MsmJob commit_job(hipStream_t st, const sonic_srs* srs, const Fr* poly, long lo, long len, long maxm, MsmSlot* slot, int* d_flags) {
  const long d = srs_d(srs), shift = d - maxm;
  long i0 = -d - shift - lo, i1 = d - shift - lo + 1;     // in-range i: lo + i + shift in [-d, d]
  if (i0 < 0) i0 = 0;
  if (i1 > len) i1 = len;
  if (i1 < i0) i1 = i0;
  flag_nonzero_enqueue(st, poly, i0, d_flags, FLAG_SRS_INDEX);
  flag_nonzero_enqueue(st, poly + i1, len - i1, d_flags, FLAG_SRS_INDEX);
  const long ih = -shift - lo;
  if (ih >= i0 && ih < i1) flag_nonzero_enqueue(st, poly + ih, 1, d_flags, FLAG_SRS_INDEX);
  return MsmJob{srs_basis(srs, 1) + (lo + i0 + shift + d), poly + i0, i1 - i0, slot};
}

// The MSMs that become ready together run as one batched kernel chain when the SRS has window tables (msm.hpp);
// otherwise one after the other.
void run_jobs(hipStream_t st, const sonic_srs* srs, MsmWorkspace& ws, const MsmJob* jobs, int k, bool last, bool exposed) {
  if (k <= 0) return;
  long nmax = 0;
  for (int j = 0; j < k; j++) nmax = std::max(nmax, jobs[j].n);
  MsmPlan pl = srs_msm_plan(srs, nmax);
  // Which groups reduce their buckets by the bit-sum butterfly: the group that finishes last, and every group when the caller knows
  // that little else runs beside it (`exposed`: a rank's share of a proof split over many GPUs); the others keep the running sums
  // over K-bucket segments.  Both forms cost the same instructions per bucket set (1.8e8 wave-instructions at 2^19 buckets:
  // profiles/r04_valu_budget.txt); the butterfly is one addition deep instead of ~2K + 30, the segments touch every bucket once.
  // Butterfly in EVERY group, measured again in round 5 (same box, alternating, ms per proof): streamed 33.76 / 33.91 against
  // 33.67 / 33.76, one at a time 34.75 / 34.98 against 34.21 / 34.50, n = 2^20 127.6 / 127.5 against 126.6 / 127.2 -- not adopted.
  // Small bucket sets (<= 2^17, round 6): always the butterfly.  Their running sums are pure latency -- K = 16 segments and the 256-lane
  // window tree took 0.95 + 0.43 ms per group at 2^16 buckets, whatever ran beside them, against 0.12 + 0.12 ms for the butterfly
  // (profiles/r06_small_proofs.txt) -- and a small proof has little accumulation to hide 1.4 ms under.
  const bool tree = last || exposed || pl.NB <= (1 << 17);
  if (k > 1 && msm_can_batch(pl)) {
    // the group that finishes last reduces with nothing left to hide under: shortest chain instead of least work
    msm_plan_set_segment(pl, prove_segment(pl, k, last));
    pl.tree = tree;
    msm_enqueue_batch(st, ws, pl, jobs, k, true);
    return;
  }
  for (int j = 0; j < k; j++) {
    MsmPlan p1 = srs_msm_plan(srs, jobs[j].n);
    msm_plan_set_segment(p1, prove_segment(p1, 1, false));
    p1.tree = tree;
    msm_enqueue_batch(st, ws, p1, &jobs[j], 1, true);
  }
}

// f(z) for a dense Laurent f: D_e = c_e z^e, inclusive prefix sums, f(z) = last prefix.
static void eval_prefix_enqueue(hipStream_t st, Scratch& sc, const Fr* poly, long lo, long len, const Fr* zpair, Fr* d_fz) {
  sc.reserve(len);
  sc.scan.ensure(sizeof(Fr) * (len / 1024 + 2));
  poly_scale_powers_enqueue(st, poly, sc.D.as<Fr>(), len, lo, zpair, zpair + 1);
  poly_prefix_sum_enqueue(st, sc.D.as<Fr>(), len, sc.scan);
  HIP_OK(hipMemcpyAsync(d_fz, sc.D.as<Fr>() + (len - 1), sizeof(Fr), hipMemcpyDeviceToDevice, st));
}

// openPoly (CommitmentScheme.hs:36-48).  Requires lo <= 0 <= lo+len-1 (callers extend the range to
// contain X^0, where `fX - monomial 0 fz` puts -f(z)).  Quotient exponents [lo, lo+len-2], plain basis.
// Queues evaluation + quotient and returns the MSM that is left to run (its scalars live in `sc` until then).
MsmJob open_job(hipStream_t st, const sonic_srs* srs, Scratch& sc, const Fr* poly, long lo, long len,
                       const Fr* zpair, Fr* d_fz, MsmSlot* slot, int* d_flags) {
  const long d = srs_d(srs);
  sc.reserve(len);
  eval_prefix_enqueue(st, sc, poly, lo, len, zpair, d_fz ? d_fz : sc.fz_discard.as<Fr>());
  const long qn = len - 1;
  poly_quotient_enqueue(st, sc.D.as<Fr>(), sc.q.as<Fr>(), len, lo, zpair, zpair + 1);
  long i0 = -d - lo, i1 = d - lo + 1;
  if (i0 < 0) i0 = 0;
  if (i1 > qn) i1 = qn;
  if (i1 < i0) i1 = i0;
  const Fr* q = sc.q.as<Fr>();
  flag_nonzero_enqueue(st, q, i0, d_flags, FLAG_SRS_INDEX);
  flag_nonzero_enqueue(st, q + i1, qn - i1, d_flags, FLAG_SRS_INDEX);
  return MsmJob{srs_basis(srs, 0) + (lo + i0 + d), q + i0, i1 - i0, slot};
}

// openPoly at z = 0 of a polynomial without negative exponents (lo == 0): f(0) = c_0 and (f - c_0)/X is the coefficient
// array shifted down by one (CommitmentScheme.hs:43-44 with z = 0).  The prefix-sum form above multiplies by z^{-1-j} and
// cannot express it (0^-1 is not defined; k_fr_with_inverse returns 0 for it).
MsmJob open_job_at_zero(hipStream_t st, const sonic_srs* srs, const Fr* poly, long len, Fr* d_fz, MsmSlot* slot, int* d_flags) {
  const long d = srs_d(srs);
  HIP_OK(hipMemcpyAsync(d_fz, poly, sizeof(Fr), hipMemcpyDeviceToDevice, st));
  const long qn = len - 1;                 // quotient exponents [0, len - 2]
  long i1 = d + 1;
  if (i1 > qn) i1 = qn;
  flag_nonzero_enqueue(st, poly + 1 + i1, qn - i1, d_flags, FLAG_SRS_INDEX);
  return MsmJob{srs_basis(srs, 0) + d, poly + 1, i1, slot};
}

// openPoly for several openings over one exponent range as one batched set of launches (poly.hip, open_batch_enqueue); the jobs
// that are left to run come back in jobs_out, in the order of `ops`
struct PendingOpen { const Fr* poly; long lo, len; const Fr* zp; Fr* fz; MsmSlot* slot; long slot_index; Scratch* sc; };
static void open_jobs_batched(hipStream_t st, const sonic_srs* srs, const PendingOpen* ops, int k, int* d_flags, MsmJob* jobs_out) {
  const long d = srs_d(srs);
  for (int at = 0; at < k;) {
    int e = at;
    OpenBatch b;
    memset(&b, 0, sizeof b);
    while (e < k && e - at < OPEN_BATCH_MAX && ops[e].lo == ops[at].lo && ops[e].len == ops[at].len) {
      Scratch& sc = *ops[e].sc;
      const long len = ops[e].len;
      sc.reserve(len);
      sc.scan.ensure(sizeof(Fr) * (len / 1024 + 2));
      const int i = e - at;
      b.poly[i] = ops[e].poly; b.D[i] = sc.D.as<Fr>(); b.q[i] = sc.q.as<Fr>(); b.tiles[i] = sc.scan.as<Fr>();
      b.zpair[i] = ops[e].zp; b.fz[i] = ops[e].fz ? ops[e].fz : sc.fz_discard.as<Fr>();
      e++;
    }
    b.k = e - at;
    const long lo = ops[at].lo, len = ops[at].len;
    open_batch_enqueue(st, b, lo, len);
    for (int i = at; i < e; i++) {
      const long qn = len - 1;
      long i0 = -d - lo, i1 = d - lo + 1;
      if (i0 < 0) i0 = 0;
      if (i1 > qn) i1 = qn;
      if (i1 < i0) i1 = i0;
      const Fr* q = ops[i].sc->q.as<Fr>();
      flag_nonzero_enqueue(st, q, i0, d_flags, FLAG_SRS_INDEX);
      flag_nonzero_enqueue(st, q + i1, qn - i1, d_flags, FLAG_SRS_INDEX);
      jobs_out[i] = MsmJob{srs_basis(srs, 0) + (lo + i0 + d), q + i0, i1 - i0, ops[i].slot};
    }
    at = e;
  }
}

bool bytes_are_zero(const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return false; return true; }

}  // namespace sonic

int upload_fr_mont(hipStream_t st, DevBuf& dst, const uint8_t* src, long count, int* d_flags) {
  dst.ensure(sizeof(Fr) * (count > 0 ? count : 1));
  if (count > 0) {
    HIP_OK(hipMemcpyAsync(dst.p, src, 32 * count, hipMemcpyHostToDevice, st));
    fr_to_mont_enqueue(st, dst.as<Fr>(), count, d_flags);
  }
  return 0;
}

int read_flags(hipStream_t st, DevBuf& flags) {
  int h = 0;
  HIP_OK(hipMemcpyAsync(&h, flags.p, 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  return h;
}

int flags_to_status(int f, const char* who) {
  if (f & FLAG_BAD_ENCODING) { set_error("%s: non-canonical field element in input", who); return SONIC_ERR_BAD_ENCODING; }
  if (f & FLAG_SRS_INDEX) { set_error("%s: a non-zero coefficient needs an SRS element outside [-d, d] or the omitted g^alpha (index -1)", who); return SONIC_ERR_SRS_INDEX; }
  return SONIC_OK;
}

// the circuit of a handle: Q x n weights and Q constants, uploaded and brought to Montgomery form (sonic_prover_new; the one-shot
// sonic_prove re-uses a cached shell by loading the next call's circuit into it)
// 32 tiles of RUN_TILE consecutive gate indices, spread over [0, n): a tile counts when every row of wL AND of wR repeats one value across
// it -- then s(X, y) has a run of equal coefficients there (u_i = sum_q wL[q][i] y^{n+q}, Constraints.hs:39-49) -- and the circuit "has
// runs" when at least a quarter of the sampled tiles do.  ~0.5 MB read at Q = 2, microseconds.
bool circuit_runs_hint(const uint8_t* wL, const uint8_t* wR, long n, long Q) {
  const long ntiles = n / RUN_TILE;
  if (ntiles < 1) return false;
  const long samples = ntiles < 32 ? ntiles : 32;
  long uniform = 0;
  for (long sidx = 0; sidx < samples; sidx++) {
    const long t = sidx * ntiles / samples;
    bool uni = true;
    for (int m = 0; m < 2 && uni; m++) {
      const uint8_t* w = m ? wR : wL;
      for (long q = 0; q < Q && uni; q++) {
        const uint8_t* row = w + 32 * (q * n + t * RUN_TILE);
        for (long i = 1; i < RUN_TILE && uni; i++) uni = memcmp(row, row + 32 * i, 32) == 0;
      }
    }
    uniform += uni ? 1 : 0;
  }
  return 4 * uniform >= samples;
}

static int prover_load_circuit(sonic_prover* p, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO, const uint8_t* cs) {
  hipStream_t st = p->st;
  const long n = p->n, Q = p->Q;
  p->circuit_has_runs = circuit_runs_hint(wL, wR, n, Q);
  HIP_OK(hipMemsetAsync(p->flags.p, 0, 4, st));
  upload_fr_mont(st, p->wL, wL, Q * n, p->flags.as<int>());
  upload_fr_mont(st, p->wR, wR, Q * n, p->flags.as<int>());
  upload_fr_mont(st, p->wO, wO, Q * n, p->flags.as<int>());
  upload_fr_mont(st, p->cs, cs, Q, p->flags.as<int>());
  int f = read_flags(st, p->flags);
  if (f) return flags_to_status(f, "sonic_prover_new");
  return SONIC_OK;
}

extern "C" {

int sonic_srs_new(int64_t d, const uint8_t x[32], const uint8_t alpha[32], sonic_srs_t** out) { return sonic_srs_new_on(-1, d, x, alpha, out); }
int sonic_srs_new_on(int device, int64_t d, const uint8_t x[32], const uint8_t alpha[32], sonic_srs_t** out) {
  API_BEGIN_ON(device)
  if (d < 1 || !x || !alpha || !out) { set_error("sonic_srs_new: bad argument"); return SONIC_ERR_INVALID_ARG; }
  Fr xs, as;
  memcpy(xs.l, x, 32); memcpy(as.l, alpha, 32);
  if (!fp_is_canonical(xs) || !fp_is_canonical(as)) { set_error("sonic_srs_new: x or alpha not < r"); return SONIC_ERR_BAD_ENCODING; }
  if (xs.is_zero()) { set_error("sonic_srs_new: x = 0 has no inverse (recip x, SRS.hs:29)"); return SONIC_ERR_INEXACT_DIVISION; }
  std::lock_guard<std::mutex> g(call_mutex());
  sonic_srs* s = srs_alloc(d);
  try { srs_generate(default_stream(), s, xs, as); } catch (...) { sonic_srs_free(s); throw; }
  srs_set_trapdoor(s, xs, as);
  *out = s;
  API_END
}

size_t sonic_proof_size(int64_t Q) { return (size_t)((7 + 4 * Q) * 96 + (5 + 2 * Q) * 32); }

int sonic_prover_new(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
                     const uint8_t* cs, sonic_prover_t** out) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || n < 1 || Q < 1 || !wL || !wR || !wO || !cs || !out) { set_error("sonic_prover_new: bad argument (need n >= 1, Q >= 1)"); return SONIC_ERR_INVALID_ARG; }
  if (srs_d(srs) < 7 * n) {                                                   // Protocol.hs:54-55
    set_error("Parameter d is not large enough: %ld should be greater than %ld", (long)srs_d(srs), (long)(7 * n));
    return SONIC_ERR_D_TOO_SMALL;
  }
  std::unique_ptr<sonic_prover> p(new sonic_prover());
  p->srs = srs; p->n = n; p->Q = Q;
  p->device = srs_device(srs);
  {
    const MsmPlan probe = srs_msm_plan(srs, 3 * n);
    p->small_plan = msm_can_batch(probe) && probe.NB <= (1 << 17);
  }
  int prio_low = 0, prio_high = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);        // (numerically lowest = highest priority)
  const char* pe = getenv("SONIC_PROVE_PRIORITIES");
  const bool use_prio = p->small_plan && pe && atoi(pe) == 1 && prio_low != prio_high;
  auto mkstream = [&](hipStream_t* s, int prio) {
    if (use_prio) HIP_OK(hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio));
    else HIP_OK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
  };
  mkstream(&p->st, prio_high);
  mkstream(&p->ts, prio_high);
  hipStream_t st = p->st;
  p->flags.alloc(8);
  int rc_c = prover_load_circuit(p.get(), wL, wR, wO, cs);
  if (rc_c) return rc_c;
  // workspaces
  const long tlen = 7 * n + 9;
  int lg = 0;
  while ((1L << lg) < tlen) lg++;
  p->log2m = lg;
  p->ntt = &device_ntt_tables(lg);
  const long M = 1L << lg;
  p->fa.alloc(sizeof(Fr) * M); p->fb.alloc(sizeof(Fr) * M);
  p->r1.alloc(sizeof(Fr) * (3 * n + 5));
  p->sy0.alloc(sizeof(Fr) * (3 * n + 1));
  p->syj.resize(Q);
  for (auto& b : p->syj) b.alloc(sizeof(Fr) * (3 * n + 1));
  p->su.alloc(sizeof(Fr) * (2 * n + Q + 1));
  p->pw.alloc(sizeof(Fr) * (3 * n + Q + 2));
  p->kpow.alloc(sizeof(Fr) * Q);
  p->S.alloc(sizeof(Fr) * (8 + 2 * Q)); p->PAIRS.alloc(sizeof(Fr) * 2 * (5 + 2 * Q));
  p->slots.alloc(sizeof(MsmSlot) * (7 + 5 * Q + 1));        // 7 + 4Q results, Q second halves of the S_j, 1 second half of C
  p->use_graph = getenv("SONIC_PROVE_GRAPH") && atoi(getenv("SONIC_PROVE_GRAPH")) != 0;
  HIP_OK(hipHostMalloc((void**)&p->h_tr, 32 * (8 + 2 * Q), hipHostMallocDefault));
  HIP_OK(hipHostMalloc((void**)&p->h_pairs, sizeof(Fr) * 2 * (5 + 2 * Q), hipHostMallocDefault));
  HIP_OK(hipHostMalloc((void**)&p->h_slots, sizeof(MsmSlot) * (7 + 5 * Q + 1), hipHostMallocDefault));
  HIP_OK(hipHostMalloc((void**)&p->h_fr, 32 * (3 + 2 * Q), hipHostMallocDefault));
  HIP_OK(hipHostMalloc((void**)&p->h_flags, 8, hipHostMallocDefault));
  p->h_flags[0] = p->h_flags[1] = 0;
  p->frout.alloc(sizeof(Fr) * (3 + 2 * Q));
  p->frstd.alloc(sizeof(Fr) * (3 + 2 * Q));
  HIP_OK(hipMemsetAsync(p->frout.p, 0, sizeof(Fr) * (3 + 2 * Q), st));
  HIP_OK(hipMemsetAsync(p->slots.p, 0, sizeof(MsmSlot) * (7 + 5 * Q + 1), st));      // W = 0: an empty sum until the slot's MSM has run
  memset(p->h_slots, 0, sizeof(MsmSlot) * (7 + 5 * Q + 1));
  auto mkev = [](hipEvent_t* e) { HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming)); };
  mkev(&p->ev_r1); mkev(&p->ev_sy0); mkev(&p->ev_t); mkev(&p->ev_su);
  p->ev_syj.resize(Q, nullptr);
  for (auto& e : p->ev_syj) mkev(&e);
  // MSM workspaces and opening scratch grow on first use: every proof maps the same group of MSMs to the same lane
  if (p->small_plan) {
    const char* le = getenv("SONIC_FUSED_LANES");
    const int v = le ? atoi(le) : N_LANES;                 // 0: no lane of its own (few_streams); k: k lanes
    p->few_streams = v <= 0;
    p->n_lanes = v < 1 ? 0 : (v > N_LANES ? N_LANES : v);
    if (p->few_streams) {
      p->main_lane.st = p->st; p->ts_lane.st = p->ts;
      mkev(&p->main_lane.done); mkev(&p->main_lane.prep);
      mkev(&p->ts_lane.done); mkev(&p->ts_lane.prep);
    }
  }
  for (int i = 0; i < p->n_lanes; i++) {
    Lane& l = p->lanes[i];
    mkstream(&l.st, prio_high);
    mkev(&l.done);
    mkev(&l.prep);
  }
  if (p->small_plan) {
    // (the second chain stream is only used by proofs of more than MSM_MAX_JOBS MSMs: Q > 2)
    for (int c = 0; c < (7 + 4 * Q > MSM_MAX_JOBS ? 2 : 1); c++) {
      mkstream(&p->chain[c].st, prio_low);
      mkev(&p->chain[c].done);
    }
  }
  HIP_OK(hipStreamSynchronize(st));
  *out = p.release();
  API_END
}

int sonic_prover_set_assignment(sonic_prover_t* p, const uint8_t* aL, const uint8_t* aR, const uint8_t* aO) {
  API_BEGIN_ON(p ? p->device : -1)
  if (!p || !aL || !aR || !aO) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(p->mu);
  if (p->in_flight) { set_error("sonic_prover_set_assignment: a submitted proof is still reading the current assignment (collect it first)"); return SONIC_ERR_INVALID_ARG; }
  hipStream_t st = p->st;
  HIP_OK(hipMemsetAsync(p->flags.p, 0, 4, st));
  upload_fr_mont(st, p->aL, aL, p->n, p->flags.as<int>());
  upload_fr_mont(st, p->aR, aR, p->n, p->flags.as<int>());
  upload_fr_mont(st, p->aO, aO, p->n, p->flags.as<int>());
  int f = read_flags(st, p->flags);
  if (f) {
    // the handle's buffers now hold a partly converted assignment: it must not be proven with (a caller that ignores the status and
    // calls prove gets "no assignment set", not a proof of garbage)
    p->have_assignment = false;
    return flags_to_status(f, "sonic_prover_set_assignment");
  }
  p->have_assignment = true;
  p->have_witness_digest = false;
  API_END
}

}  // extern "C"

// prove = prove_enqueue (queues the whole proof on the handle's streams, no host synchronisation) + prove_finish (waits, runs
// the host tails, lays out the bytes).  sonic_prover_prove runs them back to back; sonic_prover_submit / sonic_prover_collect
// expose the halves, so that ONE host thread can keep two handles busy: while it waits for and finishes proof i on one handle,
// proof i + 1 is already running on the other (its polynomial building and sorts fill the first proof's reduction tail).
// Both run under p->mu.
static int prove_enqueue(sonic_prover_t* p, const uint8_t* transcript) {
  API_BEGIN_ON(p->device)
  p->t_begin = std::chrono::steady_clock::now();
  const long n = p->n, Q = p->Q;
  const sonic_srs* srs = p->srs;
  hipStream_t st = p->st;
  // evaluation points feed `pow x e` with negative e (Utils.hs:18, poly's eval): x = 0 has no inverse
  for (long k = 4; k < 8 + 2 * Q; k++)
    if (bytes_are_zero(transcript + 32 * k, 32)) { set_error("prove: transcript element %ld is zero: Laurent evaluation at 0 divides by zero", k); return SONIC_ERR_INEXACT_DIVISION; }
  if (p->share_world > 1 && (!p->share_planned || p->share_planned_prepared != p->prepared)) {
    const MsmPlan mp = srs_msm_plan(srs, 3 * n);
    const ShareCosts costs = ShareCosts::from_env();
    p->share = share_plan(n, Q, p->prepared, p->share_world, mp.NB, mp.W, costs);
    // what the plan was computed from, as a 32-bit tag in the share header: the plan is a pure function of these, but NB and W come from
    // the rank's own SRS handle (window tables or not: free memory at SRS construction) and the cost constants from its environment --
    // ranks that disagree would report pieces that do not fit together, and sonic_proof_from_shares says so instead of "do not cover"
    {
      uint64_t h = 1469598103934665603ull;
      auto mix = [&](uint64_t v) { for (int i = 0; i < 8; i++) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } };
      auto mixd = [&](double d) { uint64_t v; memcpy(&v, &d, 8); mix(v); };
      mix((uint64_t)n); mix((uint64_t)Q); mix(p->prepared ? 1 : 0); mix((uint64_t)p->share_world); mix((uint64_t)mp.NB); mix((uint64_t)mp.W);
      mixd(costs.per_job_buckets); mixd(costs.r1); mixd(costs.sy); mixd(costs.su); mixd(costs.tprod);
      p->share_plan_tag = (int32_t)(uint32_t)(h ^ (h >> 32));
    }
    p->share_planned = true; p->share_planned_prepared = p->prepared;
  }
  if (p->share_world > 1 && p->phases != PH_ALL) { set_error("prove: a shared proof runs with a caller-supplied transcript only"); return SONIC_ERR_INVALID_ARG; }
  int* flags = p->flags.as<int>();
  memcpy(p->h_tr, transcript, 32 * (8 + 2 * Q));
  {
    // {v, v^-1} for v = y, z, yz, u, v, y_1..y_Q, z_1..z_Q (Montgomery form), on the host with one shared inversion: the same
    // values cost a proof 0.35 ms of single-thread Fermat inversions on the device before its first polynomial could be built.
    // A non-canonical element is flagged by the device's conversion of the transcript below; its pair here is then irrelevant.
    const long np = 5 + 2 * Q;
    std::vector<Fr> v((size_t)np), pre((size_t)np);
    auto tr = [&](long k) { Fr a; memcpy(a.l, transcript + 32 * k, 32); return fp_to_mont(a); };
    v[0] = tr(4); v[1] = tr(5); v[2] = fp_mul(v[0], v[1]); v[3] = tr(6 + 2 * Q); v[4] = tr(7 + 2 * Q);
    for (long t = 0; t < 2 * Q; t++) v[5 + t] = tr(6 + t);
    Fr acc = Fr::one();
    for (long i = 0; i < np; i++) { pre[i] = acc; acc = fp_mul(acc, v[i].is_zero() ? Fr::one() : v[i]); }
    Fr inv = fp_inv(acc);
    for (long i = np - 1; i >= 0; i--) {
      p->h_pairs[2 * i] = v[i];
      if (v[i].is_zero()) { p->h_pairs[2 * i + 1] = v[i]; continue; }
      p->h_pairs[2 * i + 1] = fp_mul(inv, pre[i]);
      inv = fp_mul(inv, v[i]);
    }
  }
  const int K = (int)(7 + 4 * Q);
  // runs of equal coefficients in S_j go through the running sums of the alpha basis: not for a piece of a shared proof (its term
  // ranges cut the runs), and by default from n = 2^16 (measured, unprepared ms per proof with / without: n = 2^18 32.9 / 36.7,
  // 2^17 19.8 / 20.8, 2^16 12.2 / 12.3, 2^14 6.8 / 6.0 -- below, the extra launches cost more than the additions they save;
  // profiles/r05_runs_ab.txt).  SONIC_PROVE_RUNS=0: never (the plain 3n + 1-term MSM); =1: whenever there are 8 tiles (tests).
  {
    const char* re = getenv("SONIC_PROVE_RUNS");
    const int mode = re ? atoi(re) : -1;
    const bool size_ok = mode == 1 ? 3 * p->n + 1 >= 8 * RUN_TILE : p->n >= (1L << 16);
    p->runs_on = mode != 0 && size_ok && (mode == 1 || p->circuit_has_runs) && !p->prepared && p->share_world <= 1 && srs_prefix(p->srs).p != nullptr;
  }
  {
    // by default from n = 2^17: the Q-term MSM's launches cost a small proof more than n additions save it (ms per proof streamed with /
    // without, profiles/r05_sym_ab.txt: n = 2^20 113.0-113.4 / 115.6-115.7, 2^18 31.35-31.6 / 31.95-32.0, 2^16 11.3-11.5 / 10.9-11.1,
    // 2^14 6.1-6.6 / 5.8-6.1).  SONIC_PROVE_SYM=0: never; =1: always (tests).  Not for a piece of a shared proof (the plan counts C's terms).
    const char* se = getenv("SONIC_PROVE_SYM");
    const int mode = se ? atoi(se) : -1;
    // (round 6, n = 2^16 again, where a proof is one chain now: with / without 9.20-9.25 / 9.08-9.15 ms streamed, 10.44-10.51 / 10.49-10.80
    // one at a time -- still no gain below 2^17; profiles/r06_ab_small_final.txt)
    p->sym_on = mode != 0 && (mode == 1 || p->n >= (1L << 17)) && p->share_world <= 1 && srs_sym(p->srs).p != nullptr;
  }
  {
    const MsmPlan probe = srs_msm_plan(srs, 3 * n);
    const char* fe = getenv("SONIC_PROVE_FUSED");
    const int mode = fe ? atoi(fe) : -1;
    p->fused = mode != 0 && p->small_plan && msm_can_batch(probe) && p->share_world <= 1;
    p->fused_jobs.clear();
    p->fused_sc_next = 0;
  }
  const int KS = p->sym_on ? K + (int)Q + 1 : K + ((p->prepared || p->runs_on) ? (int)Q : 0);        // + the second halves of the S_j and of C
  // Launch-bound sizes replay the whole multi-stream enqueue as one hipGraph: captured on the second proof of a handle (the
  // first one grows the workspaces), every address in it is owned by the handle.
  const bool pending = p->pend_circuit[0] != nullptr || p->pend_asg[0] != nullptr;
  const bool want_graph = p->use_graph && p->proofs_done >= 1 && !profiler().on && p->phases == PH_ALL && !pending;
  const bool replay = want_graph && p->graph != nullptr;
  const bool capturing = want_graph && !replay && !p->graph_tried;
  struct CaptureGuard {        // an error while capturing must not leave the stream in capture mode
    hipStream_t st; bool active = false;
    ~CaptureGuard() { if (active) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(st, &g); if (g) (void)hipGraphDestroy(g); } }
  } capture{st};
  if (capturing) { p->graph_tried = true; HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed)); capture.active = true; }
  if (!replay) {
  HIP_OK(hipMemsetAsync(flags, 0, 8, st));
  Fr* S = p->S.as<Fr>();
  HIP_OK(hipMemcpyAsync(S, p->h_tr, 32 * (8 + 2 * Q), hipMemcpyHostToDevice, st));
  fr_to_mont_enqueue(st, S, 8 + 2 * Q, flags);
  Fr* PR = p->PAIRS.as<Fr>();
  HIP_OK(hipMemcpyAsync(PR, p->h_pairs, sizeof(Fr) * 2 * (5 + 2 * Q), hipMemcpyHostToDevice, st));
  const Fr *pY = PR + 0, *pZ = PR + 2, *pYZ = PR + 4, *pU = PR + 6, *pV = PR + 8;
  auto pYj = [&](long j) { return PR + 2 * (5 + j); };
  auto pZj = [&](long j) { return PR + 2 * (5 + Q + j); };
  MsmSlot* slots = p->slots.as<MsmSlot>();
  Fr* frout = p->frout.as<Fr>();
  Fr *r1 = p->r1.as<Fr>(), *su = p->su.as<Fr>(), *pw = p->pw.as<Fr>(), *fa = p->fa.as<Fr>(), *fb = p->fb.as<Fr>();
  const Fr *wL = p->wL.as<Fr>(), *wR = p->wR.as<Fr>(), *wO = p->wO.as<Fr>(), *cs = p->cs.as<Fr>();
  const long d = srs_d(srs);
  const long r_lo = -2 * n - 4, r_len = 3 * n + 5, s_lo = -n, s_len = 3 * n + 1, t_lo = -4 * n - 8, t_len = 7 * n + 9;
  const long M = 1L << p->log2m;

  // The main stream builds the polynomials.  The commitments / openings that depend on one polynomial form a group:
  // it runs on the next MSM lane as soon as its input exists (event), as ONE batched MSM kernel chain.
  // Results land in disjoint slots / frout entries.
  hipStream_t ms = st;
  p->next_lane = 0;
  auto ready = [&](hipEvent_t e) { HIP_OK(hipEventRecord(e, ms)); };
  auto on = [&](int ph) { return ((p->phases >> ph) & 1u) != 0; };
  // one proof over several GPUs: this rank's pieces of the MSMs (share_plan.hpp); sh == nullptr: everything
  const SlotShare* sh = p->share_world > 1 ? p->share.row(p->share_rank) : nullptr;
  auto own = [&](long slot) { return !sh || sh[slot].hi > sh[slot].lo; };
  auto first_piece = [&](long slot) { return !sh || (sh[slot].hi > sh[slot].lo && sh[slot].lo == 0); };
  // cuts the job down to this rank's term range; false: nothing of it is left
  auto my_piece = [&](MsmJob& job, long slot) {
    if (sh) {
      long t0, t1;
      share_term_range(sh[slot], job.n, &t0, &t1);
      job.points = job.points + t0; job.scalars += t0; job.n = t1 - t0;
      if (job.n <= 0) return false;
    }
    p->slot_ran[(size_t)slot] = 1;
    return true;
  };
  p->slot_ran.assign((size_t)(7 + 5 * Q + 1), 0);
  p->fr_valid.assign((size_t)(3 + 2 * Q), 0);
  bool need_j_any = false, need_su = own(6 + 4 * Q) || own(5 + 4 * Q);
  std::vector<uint8_t> need_j((size_t)Q, 0);
  for (long j = 0; j < Q; j++) {
    need_j[(size_t)j] = own(5 + 2 * j) || own(6 + 2 * j) || own(5 + 2 * Q + 2 * j);
    need_j_any = need_j_any || need_j[(size_t)j];
    need_su = need_su || own(6 + 2 * Q + 2 * j);
  }
  const bool need_g0 = own(0) || own(2) || own(3), need_T = own(1) || own(4);
  // the group whose reduction nothing is left to hide under (the t group when this rank has a piece of it)
  long last_j = -1;
  for (long j = 0; j < Q; j++) if (need_j[(size_t)j]) last_j = j;
  const int last_group = need_T ? 3 : need_su ? 2 : need_j_any ? 1 : 0;
  Lane* cur = nullptr;
  // (packing consecutive groups into ONE batched chain -- fewer, wider chains -- was measured in round 3 and removed in round 5: n = 2^18
  // 35.9 / 38.9 ms streamed / one at a time packed against 35.0 / 35.9: the chains of one proof overlap less)
  std::vector<std::function<void()>> after_flush;       // small MSMs that use the lane's workspace after the batch (stream order)
  std::vector<std::function<void()>> deferred_small;    // (fused proofs: the same, queued behind the proof's chain)
  // the openings of the group that is being assembled: evaluated and divided together when the group is flushed
  std::vector<PendingOpen> pend;
  auto issue_opens = [&] {
    if (pend.empty()) return;
    MsmJob oj[MSM_MAX_JOBS];
    open_jobs_batched(cur->st, srs, pend.data(), (int)pend.size(), flags, oj);
    for (size_t i = 0; i < pend.size(); i++) if (my_piece(oj[i], pend[i].slot_index)) cur->jobs[cur->njobs++] = oj[i];
    pend.clear();
  };
  auto flush_now = [&](bool last = false) {
    if (!cur) return;
    issue_opens();
    if (p->fused) p->fused_jobs.insert(p->fused_jobs.end(), cur->jobs, cur->jobs + cur->njobs);      // run at the end, as one chain
    else run_jobs(cur->st, srs, cur->ws, cur->jobs, cur->njobs, last, /*exposed=*/p->share_world >= 4);
    cur->njobs = 0;
    // the small MSMs beside a group (Q-term sums): at once behind the group's chain -- or, when the proof is ONE chain, after that chain has
    // been queued: they are ~14 launches each whose results are only read at the very end, and on a stream the chain waits for they held its
    // start back by 0.8 ms (n = 2^16 streamed: profiles/r06_streamed_handover.txt)
    if (p->fused) deferred_small.insert(deferred_small.end(), after_flush.begin(), after_flush.end());
    else for (auto& f : after_flush) f();
    after_flush.clear();
  };
  // (few_streams: on_ts = the group rides on the transform's stream -- r(X,1)'s, queued there ahead of the product --, else on the main stream)
  auto begin_group = [&](hipEvent_t e, bool on_ts = false) {
    flush_now();
    if (p->few_streams && on_ts) { cur = &p->ts_lane; HIP_OK(hipStreamWaitEvent(p->ts, e, 0)); }
    else cur = &p->pick(e);
    cur->njobs = 0;
  };
  auto flush_group = [&](bool last = false) { flush_now(last); };
  auto full = [&] { return cur->njobs + (int)pend.size() == MSM_MAX_JOBS; };
  auto commit = [&](int ph, const Fr* poly, long lo, long len, long maxm, long slot) {
    if (!on(ph) || !own(slot)) return;
    if (full()) flush_now();
    MsmJob job = commit_job(cur->st, srs, poly, lo, len, maxm, &slots[slot], flags);
    if (my_piece(job, slot)) cur->jobs[cur->njobs++] = job;
  };
  // commitPoly with the runs of equal coefficients taken out (S_j of a handle that is not prepared): the flag checks read the
  // coefficients themselves, the large MSM a copy with the uniform tiles zeroed, and a small MSM over gathered running sums -- on the
  // main stream, beside this lane's batch -- adds c (ps[b] - ps[a - 1]) per run
  auto commit_runs = [&](int ph, const Fr* poly, long lo, long len, long maxm, long slot, long j) {
    if (!on(ph) || !own(slot)) return;
    if (full()) flush_now();
    MsmJob job = commit_job(cur->st, srs, poly, lo, len, maxm, &slots[slot], flags);
    const long ntiles = job.n / RUN_TILE;
    if (ntiles > 0) {
      if ((long)p->runs.size() < Q) p->runs.resize((size_t)Q);
      sonic_prover::RunBufs& rb = p->runs[(size_t)j];
      if (!rb.masked_ev) HIP_OK(hipEventCreateWithFlags(&rb.masked_ev, hipEventDisableTiming));
      rb.masked.ensure(sizeof(Fr) * job.n); rb.val.ensure(sizeof(Fr) * ntiles); rb.uniform.ensure(4 * ntiles);
      rb.scal.ensure(sizeof(Fr) * 2 * ntiles); rb.pts.ensure(sizeof(G1Affine) * 2 * ntiles);
      run_tiles_enqueue(cur->st, job.scalars, job.n, rb.masked.as<Fr>(), rb.val.as<Fr>(), rb.uniform.as<uint32_t>());
      HIP_OK(hipEventRecord(rb.masked_ev, cur->st));
      const long first = (long)((job.points.p - srs_basis(srs, 1).p) / (long)job.points.stride);
      sonic_prover::RunBufs* rbp = &rb;
      auto run_tail = [&, rbp, first, ntiles, j, ms, slots, Q] {
        HIP_OK(hipStreamWaitEvent(ms, rbp->masked_ev, 0));
        run_terms_enqueue(ms, rbp->val.as<Fr>(), rbp->uniform.as<uint32_t>(), ntiles, srs_prefix(srs) + first, first, rbp->scal.as<Fr>(), rbp->pts.as<G1Affine>());
        msm_enqueue(ms, p->runs_ws, msm_plan(2 * ntiles), PointArray::packed(rbp->pts.as<G1Affine>()), rbp->scal.as<Fr>(), 2 * ntiles, true, &slots[(7 + 4 * Q) + j]);
      };
      if (p->fused) deferred_small.push_back(run_tail); else run_tail();
      job.scalars = rb.masked.as<Fr>();
      p->slot_ran[(size_t)((7 + 4 * Q) + j)] = 1;
    }
    if (my_piece(job, slot)) cur->jobs[cur->njobs++] = job;
  };
  // fr: index of the evaluation in frout (-1: not reported); every rank with a piece of the opening computes it (the quotient
  // needs the prefix sums anyway), the rank whose piece starts at term 0 reports it
  auto open = [&](int ph, const Fr* poly, long lo, long len, const Fr* zp, long fr, long slot) {
    if (!on(ph) || !own(slot)) return;
    if (full()) flush_now();
    Scratch& sc = p->fused ? p->fused_scratch() : cur->sc[cur->njobs + (int)pend.size()];
    pend.push_back(PendingOpen{poly, lo, len, zp, fr >= 0 ? &frout[fr] : nullptr, &slots[slot], slot, &sc});
    if (fr >= 0 && first_piece(slot)) p->fr_valid[(size_t)fr] = 1;
  };
  Fr* sy = p->sy0.as<Fr>();
  // ---- all polynomials first (small kernels; queued behind a bucket accumulation they would each wait ~0.5 ms for CUs) ----
  // zkP_1: r'(X,1)                                                                   Protocol.hs:58-63
  // (a polynomial is built in the pass that first knows its challenge -- every pass when phases == PH_ALL -- and stays in the
  // handle's buffers for the later passes of sonic_prover_prove_fs: r(X,1) from the blinders, s(X,y) and t(X,y) from y, s(X,y_j)
  // from y_j, s(u,Y) from u)
  if (p->pend_asg[0]) {
    upload_fr_mont(ms, p->aL, p->pend_asg[0], n, flags + 1);
    upload_fr_mont(ms, p->aR, p->pend_asg[1], n, flags + 1);
    upload_fr_mont(ms, p->aO, p->pend_asg[2], n, flags + 1);
    p->pend_asg[0] = nullptr;
    p->have_witness_digest = false;
  }
  if ((need_g0 || need_T) && on(PH_R)) build_r1_enqueue(ms, p->aL.as<Fr>(), p->aR.as<Fr>(), p->aO.as<Fr>(), S, n, r1);
  ready(p->ev_r1);
  // the group that needs nothing but r(X,1): queued here, ahead of the other polynomials, when this call's circuit is still on the host
  bool g0_queued = false;
  auto group0 = [&] {
    if (!need_g0 || g0_queued) return;
    g0_queued = true;
    begin_group(p->ev_r1, /*on_ts=*/true);
    commit(PH_R, r1, r_lo, r_len, n, 0);                                               // R            :63
    open(PH_OPEN, r1, r_lo, r_len, pZ, 0, 2);                                          // (a, W_a)     :79
    open(PH_OPEN, r1, r_lo, r_len, pYZ, 1, 3);                                         // (b, W_b)     :80
    flush_group(last_group == 0);
  };
  if (p->few_streams && p->fused) group0();          // (its openings go on the transform's stream, ahead of the product's kernels)
  if (p->pend_circuit[0]) {
    group0();
    const uint8_t* const* c = p->pend_circuit;
    upload_fr_mont(ms, p->wL, c[0], Q * n, flags + 1);
    upload_fr_mont(ms, p->wR, c[1], Q * n, flags + 1);
    upload_fr_mont(ms, p->wO, c[2], Q * n, flags + 1);
    upload_fr_mont(ms, p->cs, c[3], Q, flags + 1);
    p->pend_circuit[0] = nullptr;
  }
  // s(X,y)                                                                           Protocol.hs:69-70
  if (need_T && on(PH_T)) {
    poly_scale_powers_enqueue(ms, nullptr, pw, 2 * n + Q + 1, -n, pY, pY + 1);       // y^e, e in [-n, n+Q]
    s_of_y_enqueue(ms, wL, wR, wO, pw, n, Q, sy);
    HIP_OK(hipMemcpyAsync(p->kpow.p, pw + (2 * n + 1), sizeof(Fr) * Q, hipMemcpyDeviceToDevice, ms));   // y^{n+1..n+Q} for k(y); pw is reused below
  }
  ready(p->ev_sy0);
  if (need_T && on(PH_T)) {
    // zkP_2: t(X,y) = r(X,1) * (r(X,y) + s(X,y)) - k(y), on its own stream          Protocol.hs:69-73, Constraints.hs:56-68
    hipStream_t ts = p->ts;
    HIP_OK(hipStreamWaitEvent(ts, p->ev_sy0, 0));          // ev_sy0 follows ev_r1 on the main stream
    t_operands_enqueue(ts, r1, r_len, r_lo, sy, s_lo - r_lo, s_len, pY, fa, fb, M);     // fa = r(X,1), fb = r(X,y) + s(X,y), zero-padded
    ntt_forward_enqueue(ts, *p->ntt, fa, p->log2m);
    ntt_forward_enqueue(ts, *p->ntt, fb, p->log2m);
    ntt_inverse_of_product_enqueue(ts, *p->ntt, fa, fb, p->log2m);
    sub_k_of_y_enqueue(ts, fa + (0 - t_lo), cs, p->kpow.as<Fr>(), Q, flags, 0);
    HIP_OK(hipEventRecord(p->ev_t, ts));
  }
  Fr* t = fa;                                                                         // exponents [t_lo, t_lo + t_len)
  // hscProve: s(X, y_j), s(u, Y)                                                     Signature.hs:41,51
  for (long j = 0; j < Q; j++) {
    if (need_j[(size_t)j] && on(PH_HSCS)) {
      poly_scale_powers_enqueue(ms, nullptr, pw, 2 * n + Q + 1, -n, pYj(j), pYj(j) + 1);
      // a prepared handle that has only a piece of S_j's diagonal part does not read s(X, y_j) itself
      if (!p->prepared || own(6 + 2 * j) || own(5 + 2 * Q + 2 * j)) s_of_y_enqueue(ms, wL, wR, wO, pw, n, Q, p->syj[j].as<Fr>());
      if (p->prepared && own(5 + 2 * j)) s_diag_part_enqueue(ms, pw, n, Q, p->diag[j].as<Fr>(), p->yq[j].as<Fr>());
    }
    ready(p->ev_syj[j]);
  }
  const long u_lo = -n, u_len = 2 * n + Q + 1;
  if (need_su && on(PH_HSCW)) {
    poly_scale_powers_enqueue(ms, nullptr, pw, 3 * n + 1, -n, pU, pU + 1);           // u^e, e in [-n, 2n]
    s_of_u_enqueue(ms, wL, wR, wO, pw, n, Q, su, p->tmp);
  }
  ready(p->ev_su);

  // ---- the MSM groups, largest first where its input allows ----
  Lane& lane_t = p->t_lane(p->ev_sy0);
  if (on(PH_OPEN) && first_piece(4)) {                                                 // s(z,y)       :83  (reported by the rank that starts W_t)
    // (few_streams: on the main stream, not behind the transform on its stream -- the evaluation needs s(X,y) only, and everything queued
    // behind the transform is on the longest dependent chain of a proof)
    Lane& le = p->few_streams ? p->main_lane : lane_t;
    Scratch& es = le.sc[MSM_MAX_JOBS - 1];
    es.reserve(s_len);
    es.scan.ensure(sizeof(Fr) * (s_len / 1024 + 2));
    OpenBatch eb;
    memset(&eb, 0, sizeof eb);
    eb.k = 1; eb.poly[0] = sy; eb.D[0] = es.D.as<Fr>(); eb.q[0] = es.q.as<Fr>(); eb.tiles[0] = es.scan.as<Fr>(); eb.zpair[0] = pZ; eb.fz[0] = &frout[2];
    open_batch_enqueue(le.st, eb, s_lo, s_len, /*quotient=*/false);
    p->fr_valid[2] = 1;
  }
  group0();
  for (long j = 0; j < Q; j++) {
    if (!need_j[(size_t)j]) continue;
    Fr* syj = p->syj[j].as<Fr>();
    begin_group(p->ev_syj[j]);
    if (p->prepared) commit(PH_HSCS, p->diag[j].as<Fr>(), n + 1, n, d, 5 + 2 * j);   // S_j (diagonal part)   Signature.hs:42
    else if (p->runs_on) commit_runs(PH_HSCS, syj, s_lo, s_len, d, 5 + 2 * j, j);    // S_j, runs through the running sums   :42
    else commit(PH_HSCS, syj, s_lo, s_len, d, 5 + 2 * j);                            // S_j                   :42
    open(PH_HSCS, syj, s_lo, s_len, pZj(j), 3 + j, 6 + 2 * j);                       // (s_j, W_j)    :43
    open(PH_HSCW, syj, s_lo, s_len, pU, -1, 5 + 2 * Q + 2 * j);                      // W'_j          :54
    flush_group(last_group == 1 && j == last_j);
    if (p->prepared && on(PH_HSCS) && first_piece(5 + 2 * j)) {                       // sum_q y_j^{n+q} C_q, Q-term MSM
      Lane* ln = cur;
      p->slot_ran[(size_t)((7 + 4 * Q) + j)] = 1;
      auto small = [&, ln, j] {
        msm_enqueue(ln->st, ln->ws, p->cq_tab.p ? msm_plan_tables(Q, CQ_TAB_C, CQ_TAB_W, Q) : msm_plan(Q),
                    PointArray::packed(p->cq_tab.p ? p->cq_tab.as<G1Affine>() : p->cq.as<G1Affine>()), p->yq[j].as<Fr>(), Q, true, &slots[(7 + 4 * Q) + j]);
      };
      if (cur->njobs == 0) small(); else after_flush.push_back(small);
    }
  }
  if (need_su) {
    begin_group(p->ev_su);
    if (p->sym_on && on(PH_HSCW) && own(6 + 4 * Q)) {                                  // C             :52, over the symmetric sums
      if (full()) flush_now();
      MsmJob job = commit_job(cur->st, srs, su, u_lo, u_len, d, &slots[6 + 4 * Q], flags);      // (the index checks of the whole range)
      job.points = srs_sym(srs) + 1; job.scalars = su + (n + 1); job.n = n; job.table_stride = d + 1;      // exponents 1 .. n: c_i (A[i] + A[-i])
      p->slot_ran[(size_t)(6 + 4 * Q)] = 1;
      cur->jobs[cur->njobs++] = job;
      p->slot_ran[(size_t)(7 + 5 * Q)] = 1;                                                     // exponents n+1 .. n+Q, on the main stream (su was built there)
      auto c_tail = [&, su, d, n, Q, slots, ms] { msm_enqueue(ms, p->runs_ws, msm_plan(Q), srs_basis(srs, 1) + (d + n + 1), su + (2 * n + 1), Q, true, &slots[7 + 5 * Q]); };
      if (p->fused) deferred_small.push_back(c_tail); else c_tail();
    } else
      commit(PH_HSCW, su, u_lo, u_len, d, 6 + 4 * Q);                                // C             :52
    for (long j = 0; j < Q; j++) open(PH_HSCW, su, u_lo, u_len, pYj(j), 3 + Q + j, 6 + 2 * Q + 2 * j);   // (s'_j, Q_j) :55
    open(PH_QV, su, u_lo, u_len, pV, -1, 5 + 4 * Q);                                 // Q_v           :63
  }
  flush_now(last_group == 2 && sh != nullptr);
  static const bool split_t = getenv("SONIC_FUSED_SPLIT_T") && atoi(getenv("SONIC_FUSED_SPLIT_T")) != 0;
  auto t_group = [&](bool own_chain) {
    if (on(PH_T) && lane_t.st != p->ts) HIP_OK(hipStreamWaitEvent(lane_t.st, p->ev_t, 0));
    cur = &lane_t; cur->njobs = 0;
    commit(PH_T, t, t_lo, t_len, d, 1);                                                // T            Protocol.hs:73
    open(PH_OPEN, t, t_lo, t_len, pZ, -1, 4);                                          // W_t          :81
    if (p->fused && own_chain) {
      issue_opens();
      if (cur->njobs > 0) {
        long nmax = 0;
        for (int j = 0; j < cur->njobs; j++) nmax = std::max(nmax, cur->jobs[j].n);
        MsmPlan pl = srs_msm_plan(srs, nmax);
        pl.tree = true;
        msm_enqueue_batch(cur->st, cur->ws, pl, cur->jobs, cur->njobs, true);
        cur->njobs = 0;
      }
    } else
      flush_group(true);
  };
  if (need_T && !(p->fused && split_t)) t_group(false);
  if (p->fused && !p->fused_jobs.empty()) {
    // the proof's chain(s): every lane has queued the openings of its groups by now; chunks of at most MSM_MAX_JOBS jobs (one chunk up to
    // Q = 2) on the two chain streams in turn so that two chunks overlap like two groups did
    for (int i = 0; i < p->n_lanes; i++) HIP_OK(hipEventRecord(p->lanes[i].prep, p->lanes[i].st));
    if (p->few_streams) { HIP_OK(hipEventRecord(p->main_lane.prep, ms)); if (need_T || g0_queued) HIP_OK(hipEventRecord(p->ts_lane.prep, p->ts)); }
    const int total = (int)p->fused_jobs.size();
    const int nchunks = (total + MSM_MAX_JOBS - 1) / MSM_MAX_JOBS, per = (total + nchunks - 1) / nchunks;
    for (int c = 0, at = 0; c < nchunks; c++, at += per) {
      Lane& cl = p->chain[c & 1];
      if (c < 2) {
        for (int i = 0; i < p->n_lanes; i++) HIP_OK(hipStreamWaitEvent(cl.st, p->lanes[i].prep, 0));
        if (p->few_streams) { HIP_OK(hipStreamWaitEvent(cl.st, p->main_lane.prep, 0)); if (need_T || g0_queued) HIP_OK(hipStreamWaitEvent(cl.st, p->ts_lane.prep, 0)); }
      }
      const int k = std::min(per, total - at);
      long nmax = 0;
      for (int j = 0; j < k; j++) nmax = std::max(nmax, p->fused_jobs[(size_t)(at + j)].n);
      MsmPlan pl = srs_msm_plan(srs, nmax);
      pl.tree = true;
      pl.accum_block = 128;      // (measured for this chain, n = 2^16 streamed: 128 lanes 9.19-9.27 ms, 256 9.46-9.49, 512 9.42-9.46)
      msm_enqueue_batch(cl.st, cl.ws, pl, &p->fused_jobs[(size_t)at], k, true);
    }
    p->fused_jobs.clear();
  }
  // (SONIC_FUSED_SPLIT_T=1, measured and not the default: T and W_t -- 14n of a proof's 45n terms, and the group whose polynomial is ready
  // last -- as a SECOND chain on the stream that made the polynomial, so that the chain of the other thirteen MSMs need not wait for the
  // product.  One at a time it gains 1-3 %; streamed it loses 3-10 % at n = 2^14 .. 2^16: two sorts, two accumulation tails and two
  // butterflies per proof; profiles/r06_ab_small_final.txt)
  if (need_T && p->fused && split_t) t_group(true);
  for (auto& f : deferred_small) f();
  deferred_small.clear();
  for (int i = 0; i < p->n_lanes; i++) { Lane& l = p->lanes[i]; HIP_OK(hipEventRecord(l.done, l.st)); HIP_OK(hipStreamWaitEvent(ms, l.done, 0)); }
  if (p->fused) for (auto& l : p->chain) if (l.st) { HIP_OK(hipEventRecord(l.done, l.st)); HIP_OK(hipStreamWaitEvent(ms, l.done, 0)); }
  if (p->few_streams && (need_T || g0_queued)) { HIP_OK(hipEventRecord(p->ts_lane.done, p->ts_lane.st)); HIP_OK(hipStreamWaitEvent(ms, p->ts_lane.done, 0)); }
  Fr* frstd = p->frstd.as<Fr>();
  HIP_OK(hipMemcpyAsync(frstd, frout, sizeof(Fr) * (3 + 2 * Q), hipMemcpyDeviceToDevice, ms));
  fr_from_mont_enqueue(ms, frstd, 3 + 2 * Q);
  HIP_OK(hipMemcpyAsync(p->h_slots, slots, sizeof(MsmSlot) * KS, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(p->h_fr, frstd, 32 * (3 + 2 * Q), hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(p->h_flags, flags, 8, hipMemcpyDeviceToHost, st));
  }  // !replay
  if (capturing) {
    hipGraph_t g = nullptr;
    capture.active = false;
    HIP_OK(hipStreamEndCapture(st, &g));
    hipError_t ge = hipGraphInstantiate(&p->graph, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ge != hipSuccess) { p->graph = nullptr; set_error("prove: hipGraphInstantiate failed: %s", hipGetErrorString(ge)); return SONIC_ERR_HIP; }
  }
  if (replay || capturing) HIP_OK(hipGraphLaunch(p->graph, st));
  p->t_enq = std::chrono::steady_clock::now();
  API_END
}

// canonical proof bytes from the 7 + 4Q points (slot order) and the 3 + 2Q evaluations: record order of `Proof` (Protocol.hs:28-38)
// then `HscProof` (Signature.hs:22-29)
static void proof_layout(long Q, const uint8_t* pts, const uint8_t* frs, const uint8_t* transcript, uint8_t* out_proof) {
  auto G = [&](long i) { return pts + 96 * (size_t)i; };
  auto F = [&](long i) { return frs + 32 * (size_t)i; };
  uint8_t* o = out_proof;
  auto putG = [&](long i) { memcpy(o, G(i), 96); o += 96; };
  auto putF = [&](const uint8_t* s) { memcpy(o, s, 32); o += 32; };
  putG(0); putG(1); putF(F(0)); putG(2); putF(F(1)); putG(3); putG(4); putF(F(2));      // R T a Wa b Wb Wt s
  for (long j = 0; j < Q; j++) { putG(5 + 2 * j); putF(F(3 + j)); putG(6 + 2 * j); }     // hscS
  for (long j = 0; j < Q; j++) { putF(F(3 + Q + j)); putG(5 + 2 * Q + 2 * j); putG(6 + 2 * Q + 2 * j); }   // hscW
  putG(5 + 4 * Q); putG(6 + 4 * Q);                                                      // Qv, C
  putF(transcript + 32 * (6 + 2 * Q)); putF(transcript + 32 * (7 + 2 * Q));              // u, v
}

// ---- one proof over several GPUs: the share a rank reports and how the shares become the proof ---------------------------------
// share = header (32 B) | K x {lo, hi} pieces (8 B each) | K x 192-B un-normalised partial sums (infinity where the rank has no
// piece) | (3 + 2Q) x 32-B evaluations a, b, s, s_j, s'_j (standard form; zeros unless reported) | (3 + 2Q) x int32 reported
namespace {
struct ShareHeader { uint32_t magic, version; int32_t rank, world; int64_t Q; int32_t flags, plan_tag; };
constexpr uint32_t SHARE_VERSION = 2;      // 2 (round 5): plan_tag in what was padding
constexpr uint32_t SHARE_MAGIC = 0x48534e53u;      // "SNSH"
}
extern "C" size_t sonic_proof_share_size(int64_t Q) {
  const size_t K = (size_t)(7 + 4 * Q), F = (size_t)(3 + 2 * Q);
  return sizeof(ShareHeader) + K * 8 + K * 192 + F * 32 + F * 4;
}

static int prove_finish_share(sonic_prover_t* p, uint8_t* out_share) {
  API_BEGIN_ON(p->device)
  const long Q = p->Q;
  const int K = (int)(7 + 4 * Q), F = (int)(3 + 2 * Q);
  HIP_OK(hipStreamSynchronize(p->st));
  p->proofs_done++;
  if (getenv("SONIC_DEBUG_TIMING"))
    fprintf(stderr, "[sonic] share %d/%d: enqueue %.3f ms, then waited %.3f ms for the device\n", p->share_rank, p->share_world,
            std::chrono::duration<double, std::milli>(p->t_enq - p->t_begin).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p->t_enq).count());
  const MsmSlot* hs = p->h_slots;
  memset(out_share, 0, sonic_proof_share_size(Q));
  ShareHeader h{SHARE_MAGIC, SHARE_VERSION, p->share_rank, p->share_world > 1 ? p->share_world : 1, Q, *p->h_flags, p->share_world > 1 ? p->share_plan_tag : 0};
  memcpy(out_share, &h, sizeof h);
  uint8_t* o = out_share + sizeof h;
  for (int i = 0; i < K; i++) {
    SlotShare s; s.lo = 0; s.hi = SHARE_ONE;
    if (p->share_world > 1) s = p->share.row(p->share_rank)[i];
    memcpy(o, &s.lo, 4); memcpy(o + 4, &s.hi, 4); o += 8;
  }
  for (int i = 0; i < K; i++, o += 192) {
    G1XYZZ sum = G1XYZZ::inf();
    if (p->slot_ran[(size_t)i]) sum = msm_finish_host(hs[i]);
    const int j = (i - 5) / 2;
    if ((p->prepared || p->runs_on) && i >= 5 && i < 5 + 2 * Q && ((i - 5) & 1) == 0 && p->slot_ran[(size_t)(K + j)]) sum = g1_add(sum, msm_finish_host(hs[K + j]));
    if (p->sym_on && i == 6 + 4 * Q && p->slot_ran[(size_t)(K + Q)]) sum = g1_add(sum, msm_finish_host(hs[K + Q]));      // C's Q-term half
    memcpy(o, &sum, 192);
  }
  for (int i = 0; i < F; i++, o += 32) if (p->fr_valid[(size_t)i]) memcpy(o, p->h_fr + 32 * i, 32);
  for (int i = 0; i < F; i++, o += 4) { const int32_t v = p->fr_valid[(size_t)i]; memcpy(o, &v, 4); }
  API_END
}

// the shares of all ranks -> proof bytes.  Checks that the pieces of every MSM partition its term range and that every evaluation
// was reported; a rank's error flags (non-canonical input, SRS index, unsatisfied circuit) become the status, as in sonic_prove.
extern "C" int sonic_proof_from_shares(int64_t Q, int world, const uint8_t* shares, const uint8_t* transcript, uint8_t* out_proof) {
  try {
  if (Q < 1 || world < 1 || !shares || !transcript || !out_proof) return SONIC_ERR_INVALID_ARG;
  const int K = (int)(7 + 4 * Q), F = (int)(3 + 2 * Q);
  const size_t sz = sonic_proof_share_size(Q);
  std::vector<const uint8_t*> by_rank((size_t)world, nullptr);
  int flags = 0;
  int32_t tag0 = 0;
  for (int r = 0; r < world; r++) {
    ShareHeader h;
    memcpy(&h, shares + sz * r, sizeof h);
    if (r == 0) tag0 = h.plan_tag;
    if (h.magic == SHARE_MAGIC && h.version == SHARE_VERSION && h.plan_tag != tag0) {
      set_error("sonic_proof_from_shares: the ranks planned the proof with different parameters (share %d: tag %08x, share 0: %08x) -- their SRS handles "
                "run different MSM plans (window tables built on one GPU and not on another?) or their SONIC_SHARE_COST_* environments differ", r, (unsigned)h.plan_tag, (unsigned)tag0);
      return SONIC_ERR_INVALID_ARG;
    }
    if (h.magic != SHARE_MAGIC || h.version != SHARE_VERSION || h.Q != Q || h.world != world || h.rank < 0 || h.rank >= world || by_rank[(size_t)h.rank]) {
      set_error("sonic_proof_from_shares: share %d is not one of %d distinct shares of a Q = %ld proof", r, world, (long)Q); return SONIC_ERR_INVALID_ARG; }
    by_rank[(size_t)h.rank] = shares + sz * r;
    flags |= h.flags;
  }
  if (flags) return flags_to_status(flags, "prove");
  std::vector<G1XYZZ> sums((size_t)K, G1XYZZ::inf());
  for (int i = 0; i < K; i++) {
    std::vector<std::pair<uint32_t, uint32_t>> pieces;
    for (int r = 0; r < world; r++) {
      const uint8_t* b = by_rank[(size_t)r] + sizeof(ShareHeader);
      uint32_t lo, hi;
      memcpy(&lo, b + 8 * i, 4); memcpy(&hi, b + 8 * i + 4, 4);
      if (hi <= lo) continue;
      pieces.push_back({lo, hi});
      G1XYZZ part;
      memcpy(&part, b + 8 * K + 192 * (size_t)i, 192);
      sums[(size_t)i] = g1_add(sums[(size_t)i], part);
    }
    std::sort(pieces.begin(), pieces.end());
    uint32_t at = 0;
    for (auto& pc : pieces) { if (pc.first != at) break; at = pc.second; }
    if (at != SHARE_ONE) { set_error("sonic_proof_from_shares: the pieces of MSM %d do not cover its terms exactly once", i); return SONIC_ERR_INVALID_ARG; }
  }
  std::vector<uint8_t> pts(96 * (size_t)K), frs(32 * (size_t)F);
  g1_canonical_bytes_host_batch(sums.data(), K, pts.data());
  for (int i = 0; i < F; i++) {
    bool got = false;
    for (int r = 0; r < world && !got; r++) {
      const uint8_t* b = by_rank[(size_t)r] + sizeof(ShareHeader) + 8 * (size_t)K + 192 * (size_t)K;
      int32_t v; memcpy(&v, b + 32 * (size_t)F + 4 * (size_t)i, 4);
      if (v) { memcpy(&frs[32 * (size_t)i], b + 32 * (size_t)i, 32); got = true; }
    }
    if (!got) { set_error("sonic_proof_from_shares: no share reports evaluation %d", i); return SONIC_ERR_INVALID_ARG; }
  }
  proof_layout(Q, pts.data(), frs.data(), transcript, out_proof);
  } catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; }
  return SONIC_OK;
}

static int prove_finish(sonic_prover_t* p, uint8_t* out_proof) {
  API_BEGIN_ON(p->device)
  const long Q = p->Q;
  const int K = (int)(7 + 4 * Q);
  hipStream_t st = p->st;
  const uint8_t* transcript = p->h_tr;
  const auto t_begin = p->t_begin, t_enq = p->t_enq;
  const bool timing = getenv("SONIC_DEBUG_TIMING") != nullptr;
  HIP_OK(hipStreamSynchronize(st));
  p->proofs_done++;
  const MsmSlot* hs = p->h_slots;
  const uint8_t* hfr_p = p->h_fr;
  const int hflags = *p->h_flags;
  if (timing) fprintf(stderr, "[sonic] prove: enqueue %.3f ms, then waited %.3f ms for the device\n",
                      std::chrono::duration<double, std::milli>(t_enq - t_begin).count(),
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq).count());
  if (p->h_flags[1]) { p->have_assignment = false; return flags_to_status(p->h_flags[1], "prove (circuit / assignment handed over with the call)"); }
  if (hflags) return flags_to_status(hflags, "prove");
  std::vector<uint8_t> pts(96 * (size_t)K);
  {
    // host tails: with window tables a slot holds ONE window sum and there is nothing to fold; without them the Horner walks
    // over <= 64 window sums (255 doublings each) run on threads.  One shared inversion normalises all results.
    auto t0 = std::chrono::steady_clock::now();
    std::vector<G1XYZZ> sums((size_t)K);
    // S_j = slot 5 + 2j + slot K + j on a prepared handle (sum_q y_j^{n+q} C_q) and on one that takes the runs of equal coefficients
    // out (their small MSM over gathered running sums: per-window sums, a Horner walk).  Slot K + j persists over the passes of
    // sonic_prover_prove_fs: not tied to this pass's slot_ran; a slot whose MSM never ran is W = 0, the empty sum.
    const bool has_extra = p->prepared || p->runs_on;
    std::vector<G1XYZZ> extra((size_t)(has_extra ? Q : 0));
    auto is_folded = [&](int i) { return hs[i].W <= 1 || hs[i].pad1 != 0; };
    bool main_folded = true, extra_folded = true;
    for (int i = 0; i < K; i++) main_folded = main_folded && is_folded(i);
    for (int j = 0; has_extra && j < (int)Q; j++) extra_folded = extra_folded && is_folded(K + j);
    ThreadGroup th;
    G1XYZZ extra_c = G1XYZZ::inf();                        // the Q-term half of C (per-window sums: a Horner walk, on a thread of its own)
    if (p->sym_on) th.emplace_back([&] { extra_c = msm_finish_host(hs[K + Q]); });
    if (has_extra && !extra_folded) {                      // the Horner walks of the extra slots beside the main thread's tails
      const int nt = (int)std::min<long>(Q, 8);
      for (int w = 0; w < nt; w++) th.emplace_back([&, w] { for (long j = w; j < Q; j += nt) extra[(size_t)j] = msm_finish_host(hs[K + j]); });
    } else if (has_extra) {
      for (long j = 0; j < Q; j++) extra[(size_t)j] = msm_finish_host(hs[K + j]);
    }
    if (main_folded) {
      for (int i = 0; i < K; i++) sums[i] = msm_finish_host(hs[i]);
    } else {
      ThreadGroup th2;
      const int nt = K < 16 ? K : 16;
      for (int w = 0; w < nt; w++) th2.emplace_back([&, w] { for (int i = w; i < K; i += nt) sums[i] = msm_finish_host(hs[i]); });
      for (auto& x : th2) x.join();
    }
    for (auto& x : th) x.join();
    for (long j = 0; has_extra && j < Q; j++) sums[(size_t)(5 + 2 * j)] = g1_add(sums[(size_t)(5 + 2 * j)], extra[(size_t)j]);
    if (p->sym_on) sums[(size_t)(6 + 4 * Q)] = g1_add(sums[(size_t)(6 + 4 * Q)], extra_c);
    g1_canonical_bytes_host_batch(sums.data(), K, pts.data());
    if (timing) fprintf(stderr, "[sonic] host tails of %d MSMs: %.3f ms\n", K, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  proof_layout(Q, pts.data(), hfr_p, transcript, out_proof);
  API_END
}

extern "C" {

static int whole_proof_only(sonic_prover_t* p, const char* who) {
  if (p->share_world > 1) { set_error("%s: the handle runs one rank's share of a proof (sonic_prover_set_share): use collect_share / prove_share", who); return SONIC_ERR_INVALID_ARG; }
  return SONIC_OK;
}

static int prove_args_ok(sonic_prover_t* p, const char* who) {
  if (!p->have_assignment) { set_error("%s: no assignment set", who); return SONIC_ERR_INVALID_ARG; }
  if (p->in_flight) { set_error("%s: a submitted proof has not been collected yet", who); return SONIC_ERR_INVALID_ARG; }
  return SONIC_OK;
}

int sonic_prover_prove(sonic_prover_t* p, const uint8_t* transcript, uint8_t* out_proof) {
  if (!p || !transcript || !out_proof) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(p->mu);
  int rc = prove_args_ok(p, "sonic_prover_prove");
  if (!rc) rc = whole_proof_only(p, "sonic_prover_prove");
  if (!rc) rc = prove_enqueue(p, transcript);
  if (!rc) rc = prove_finish(p, out_proof);
  else if (p->st) (void)hipStreamSynchronize(p->st);      // an enqueue that failed half way: let what was queued drain
  return rc;
}

// prove with the assignment of THIS call still in the caller's host buffers (uploaded inside the proof's queue: pend_asg)
int prove_with_assignment(sonic_prover_t* p, const uint8_t* aL, const uint8_t* aR, const uint8_t* aO, const uint8_t* transcript, uint8_t* out_proof) {
  std::lock_guard<std::mutex> g(p->mu);
  if (p->in_flight) { set_error("prove: a submitted proof has not been collected yet"); return SONIC_ERR_INVALID_ARG; }
  int rc = whole_proof_only(p, "prove");
  if (rc) return rc;
  p->pend_asg[1] = aR; p->pend_asg[2] = aO; p->pend_asg[0] = aL;
  rc = prove_enqueue(p, transcript);
  if (!rc) rc = prove_finish(p, out_proof);
  else if (p->st) (void)hipStreamSynchronize(p->st);
  p->pend_asg[0] = nullptr;
  p->have_assignment = rc == SONIC_OK;          // (a failed call may have left a partly converted assignment behind)
  return rc;
}

int sonic_prover_submit(sonic_prover_t* p, const uint8_t* transcript) {
  if (!p || !transcript) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(p->mu);
  int rc = prove_args_ok(p, "sonic_prover_submit");
  if (!rc) rc = prove_enqueue(p, transcript);
  if (!rc) p->in_flight = true;
  else if (p->st) (void)hipStreamSynchronize(p->st);
  return rc;
}

int sonic_prover_collect(sonic_prover_t* p, uint8_t* out_proof) {
  if (!p || !out_proof) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(p->mu);
  if (!p->in_flight) { set_error("sonic_prover_collect: nothing was submitted"); return SONIC_ERR_INVALID_ARG; }
  if (whole_proof_only(p, "sonic_prover_collect")) return SONIC_ERR_INVALID_ARG;
  p->in_flight = false;
  return prove_finish(p, out_proof);
}

// ONE proof over `world` GPUs (SURVEY 8e "MSM-level parallelism"; BASELINE configs[3] as one n = 2^20 instance on 8 GPUs): every rank
// makes a handle for the same circuit and assignment, calls set_share(rank, world), and then submit + collect_share (or
// prove_share) with the same transcript; the shares are all-gathered (a few KB) and sonic_proof_from_shares turns them into the
// proof on every rank.  world <= 1 restores the whole proof.
int sonic_prover_set_share(sonic_prover_t* p, int rank, int world) {
  API_BEGIN_ON(p ? p->device : -1)
  if (!p || (world > 1 && (rank < 0 || rank >= world))) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(p->mu);
  if (p->in_flight) { set_error("sonic_prover_set_share: a submitted proof has not been collected yet"); return SONIC_ERR_INVALID_ARG; }
  if (p->graph) { (void)hipGraphExecDestroy(p->graph); p->graph = nullptr; }
  p->graph_tried = false;
  p->share_rank = world > 1 ? rank : 0;
  p->share_world = world > 1 ? world : 0;
  p->share_planned = false;
  // slots and evaluations of pieces this rank no longer runs must not survive from an earlier proof
  HIP_OK(hipMemsetAsync(p->slots.p, 0, sizeof(MsmSlot) * (7 + 5 * p->Q), p->st));
  HIP_OK(hipMemsetAsync(p->frout.p, 0, sizeof(Fr) * (3 + 2 * p->Q), p->st));
  HIP_OK(hipStreamSynchronize(p->st));
  API_END
}

int sonic_prover_collect_share(sonic_prover_t* p, uint8_t* out_share) {
  if (!p || !out_share) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(p->mu);
  if (!p->in_flight) { set_error("sonic_prover_collect_share: nothing was submitted"); return SONIC_ERR_INVALID_ARG; }
  p->in_flight = false;
  return prove_finish_share(p, out_share);
}

int sonic_prover_prove_share(sonic_prover_t* p, const uint8_t* transcript, uint8_t* out_share) {
  if (!p || !transcript || !out_share) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(p->mu);
  int rc = prove_args_ok(p, "sonic_prover_prove_share");
  if (!rc) rc = prove_enqueue(p, transcript);
  if (!rc) rc = prove_finish_share(p, out_share);
  else if (p->st) (void)hipStreamSynchronize(p->st);
  return rc;
}

// what the plan gives a rank (host only): K = 7 + 4Q pieces {lo, hi} in units of 1 / 2^20 of each MSM's terms (slot order R, T,
// W_a, W_b, W_t, [S_j, W_j]_j, [W'_j, Q_j]_j, Q_v, C) and the modelled cost in MSM terms.  nb, w: buckets per set and windows of
// the MSM plan (sonic_msm_plan; 0, 0: the defaults of an SRS with window tables, 2^19 and 13).
int sonic_prove_share_plan(int64_t n, int64_t Q, int prepared, int world, int rank, int64_t nb, int w, uint32_t* out_lo_hi, double* out_cost) {
  if (n < 1 || Q < 1 || world < 1 || rank < 0 || rank >= world || !out_lo_hi) return SONIC_ERR_INVALID_ARG;
  const SharePlan pl = share_plan(n, Q, prepared != 0, world, nb > 0 ? nb : (1L << 19), w > 0 ? w : 13, ShareCosts::from_env());
  for (int i = 0; i < pl.K; i++) { out_lo_hi[2 * i] = pl.row(rank)[i].lo; out_lo_hi[2 * i + 1] = pl.row(rank)[i].hi; }
  if (out_cost) *out_cost = pl.cost[(size_t)rank];
  return SONIC_OK;
}

// ---- opt-in Fiat-Shamir transcript (fs.hpp) ----------------------------------------------------------------------
// The challenges of the reference are `rnd` draws made AFTER certain proof elements exist; as hashes of those elements they
// serialise the proof: R -> y -> T -> z -> openings -> y_j, z_j -> S_j.. -> u -> C.. -> v -> Q_v.  sonic_prover_prove_fs walks
// that chain in six passes over the same enqueue (prove_enqueue with one phase bit each): every MSM of the proof still runs
// exactly once and every polynomial is built once, in the pass that first knows its challenge (round 4; it was rebuilt in every pass
// before); between passes the host waits, reads the new elements' canonical bytes and hashes.  The caller-supplied transcript stays the default and fast path (one pass, no waits).
int sonic_fs_circuit_digest(int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO, const uint8_t* cs, uint8_t out[32]) {
  if (n < 1 || Q < 1 || !wL || !wR || !wO || !cs || !out) return SONIC_ERR_INVALID_ARG;
  Sha256 h;
  h.update("sonic-hip/circuit/v1", 20);
  FsTranscript::le64(h, n); FsTranscript::le64(h, Q);
  h.update(wL, (size_t)(32 * Q * n)); h.update(wR, (size_t)(32 * Q * n)); h.update(wO, (size_t)(32 * Q * n)); h.update(cs, (size_t)(32 * Q));
  h.finish(out);
  return SONIC_OK;
}

int sonic_fs_challenges_v2(int64_t n, int64_t Q, int64_t d, const uint8_t circuit_digest[32], const uint8_t srs_id[32], const uint8_t* proof, uint8_t* out) {
  if (n < 1 || Q < 1 || d < 1 || !circuit_digest || !srs_id || !proof || !out) return SONIC_ERR_INVALID_ARG;
  fs_challenges_of_proof(n, Q, d, circuit_digest, srs_id, proof, out);
  return SONIC_OK;
}
// retired with round 3's prototype: round 4 put srs_id in the middle of the argument list under the same name, so a caller built against
// the older header passed its proof pointer where the srs id is read
int sonic_fs_challenges(int64_t, int64_t, int64_t, const uint8_t*, const uint8_t*, uint8_t*) {
  set_error("sonic_fs_challenges is retired (the transcript binds the SRS since round 4): use sonic_fs_challenges_v2, which takes the srs id");
  return SONIC_ERR_INVALID_ARG;
}

int sonic_prover_prove_fs(sonic_prover_t* p, const uint8_t circuit_digest[32], const uint8_t blinder_seed[32], uint8_t* out_proof,
                          uint8_t* out_transcript) {
  if (!p || !circuit_digest || !blinder_seed || !out_proof) return SONIC_ERR_INVALID_ARG;
  API_BEGIN_ON(p->device)
  std::lock_guard<std::mutex> g(p->mu);
  int rc = prove_args_ok(p, "sonic_prover_prove_fs");
  if (!rc) rc = whole_proof_only(p, "sonic_prover_prove_fs");
  if (rc) return rc;
  const long n = p->n, Q = p->Q, d = srs_d(p->srs);
  std::vector<uint8_t> tr(32 * (size_t)(8 + 2 * Q), 0), pf(sonic_proof_size(Q));
  uint8_t srs_id[32];
  if ((rc = sonic_fs_srs_id(p->srs, srs_id))) return rc;
  if (!p->have_witness_digest) {
    // SHA-256 of the assignment's canonical bytes, once per set_assignment: the device copy is Montgomery, so a scratch copy is
    // converted back and brought to the host
    try {
      DevBuf tmp(sizeof(Fr) * 3 * n);
      Fr* t3 = tmp.as<Fr>();
      HIP_OK(hipMemcpyAsync(t3, p->aL.p, sizeof(Fr) * n, hipMemcpyDeviceToDevice, p->st));
      HIP_OK(hipMemcpyAsync(t3 + n, p->aR.p, sizeof(Fr) * n, hipMemcpyDeviceToDevice, p->st));
      HIP_OK(hipMemcpyAsync(t3 + 2 * n, p->aO.p, sizeof(Fr) * n, hipMemcpyDeviceToDevice, p->st));
      fr_from_mont_enqueue(p->st, t3, 3 * n);
      std::vector<uint8_t> host(96 * (size_t)n);
      HIP_OK(hipMemcpyAsync(host.data(), t3, host.size(), hipMemcpyDeviceToHost, p->st));
      HIP_OK(hipStreamSynchronize(p->st));
      Sha256 h;
      h.update("sonic-hip/witness/v1", 20);
      h.update(host.data(), host.size());
      h.finish(p->witness_digest);
      p->have_witness_digest = true;
    } catch (const HipFail& f) { return f.code; }
  }
  for (long k = 0; k < 4; k++) fs_blinder(blinder_seed, circuit_digest, srs_id, p->witness_digest, (uint32_t)k, &tr[32 * k]);
  for (long k = 4; k < 8 + 2 * Q; k++) tr[32 * k] = 1;               // not drawn yet: any invertible value (results that use it are not read)
  FsTranscript t;
  t.init(n, Q, d, circuit_digest, srs_id);
  auto pass = [&](int ph) {
    p->phases = 1u << ph;
    int r = prove_enqueue(p, tr.data());
    if (!r) r = prove_finish(p, pf.data());
    else if (p->st) (void)hipStreamSynchronize(p->st);
    p->phases = PH_ALL;
    return r;
  };
  const uint8_t* R = pf.data(), * T = R + 96, * open = R + 192, * hscS = R + 576, * hscW = hscS + Q * 224, * Cc = hscW + Q * 224 + 96;
  if ((rc = pass(PH_R))) return rc;
  t.absorb("R", R, 96);
  t.challenge("y", 0, &tr[32 * 4]);
  if ((rc = pass(PH_T))) return rc;
  t.absorb("T", T, 96);
  t.challenge("z", 0, &tr[32 * 5]);
  if ((rc = pass(PH_OPEN))) return rc;
  t.absorb("open", open, 384);
  for (long j = 0; j < Q; j++) { t.challenge("yj", (uint32_t)j, &tr[32 * (6 + j)]); t.challenge("zj", (uint32_t)j, &tr[32 * (6 + Q + j)]); }
  if ((rc = pass(PH_HSCS))) return rc;
  t.absorb("hscS", hscS, (size_t)Q * 224);
  t.challenge("u", 0, &tr[32 * (6 + 2 * Q)]);
  if ((rc = pass(PH_HSCW))) return rc;
  {
    std::string w(reinterpret_cast<const char*>(Cc), 96);
    w.append(reinterpret_cast<const char*>(hscW), (size_t)Q * 224);
    t.absorb("hscW", reinterpret_cast<const uint8_t*>(w.data()), w.size());
  }
  t.challenge("v", 0, &tr[32 * (7 + 2 * Q)]);
  if ((rc = pass(PH_QV))) return rc;
  memcpy(out_proof, pf.data(), pf.size());
  if (out_transcript) memcpy(out_transcript, tr.data(), tr.size());
  API_END
}

// Circuit-only precomputation for handles that prove more than once: C_q = Commit(d, P_q), P_q the q-th constraint's
// weight polynomial.  Afterwards S_j = Commit(d, s(X, y_j)) (Signature.hs:42) is assembled as
// sum_q y_j^{n+q} C_q + Commit(d, diagonal part): the same group element from an n-term MSM instead of a 3n-term one.
int sonic_prover_prepare(sonic_prover_t* p) {
  API_BEGIN_ON(p ? p->device : -1)
  if (!p) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(p->mu);
  if (p->prepared) return SONIC_OK;
  if (p->in_flight) { set_error("sonic_prover_prepare: a submitted proof has not been collected yet"); return SONIC_ERR_INVALID_ARG; }
  if (p->graph) { (void)hipGraphExecDestroy(p->graph); p->graph = nullptr; }     // a captured proof would not know the prepared path
  p->graph_tried = false;
  const long n = p->n, Q = p->Q, d = srs_d(p->srs);
  int* flags = p->flags.as<int>();
  HIP_OK(hipMemsetAsync(flags, 0, 4, p->st));
  HIP_OK(hipStreamSynchronize(p->st));
  DevBuf slots(sizeof(MsmSlot) * Q);
  std::vector<DevBuf> rows(std::min<long>(Q, N_LANES));
  for (auto& b : rows) b.alloc(sizeof(Fr) * (3 * n + 1));
  for (long q = 0; q < Q; q++) {
    Lane& l = p->lane_at((int)(q % N_LANES));
    Fr* row = rows[q % N_LANES].as<Fr>();
    weight_row_poly_enqueue(l.st, p->wL.as<Fr>(), p->wR.as<Fr>(), p->wO.as<Fr>(), n, q, row);
    MsmJob job = commit_job(l.st, p->srs, row, -n, 3 * n + 1, d, slots.as<MsmSlot>() + q, flags);
    run_jobs(l.st, p->srs, l.ws, &job, 1);
  }
  for (int i = 0; i < p->n_lanes; i++) HIP_OK(hipStreamSynchronize(p->lanes[i].st));
  if (p->few_streams) { HIP_OK(hipStreamSynchronize(p->st)); HIP_OK(hipStreamSynchronize(p->ts)); }
  std::vector<MsmSlot> hs(Q);
  HIP_OK(hipMemcpy(hs.data(), slots.p, sizeof(MsmSlot) * Q, hipMemcpyDeviceToHost));
  int hflags = 0;
  HIP_OK(hipMemcpy(&hflags, flags, 4, hipMemcpyDeviceToHost));
  if (hflags) return flags_to_status(hflags, "sonic_prover_prepare");
  std::vector<G1Affine> cq(Q);
  {
    ThreadGroup th;
    const int nt = (int)std::min<long>(Q, 16);
    for (int w = 0; w < nt; w++)
      th.emplace_back([&, w] { for (long q = w; q < Q; q += nt) cq[q] = g1_to_affine(msm_finish_host(hs[q])); });
    for (auto& x : th) x.join();
  }
  p->cq.alloc(sizeof(G1Affine) * Q);
  HIP_OK(hipMemcpy(p->cq.p, cq.data(), sizeof(G1Affine) * Q, hipMemcpyHostToDevice));
  if (Q <= CQ_TAB_MAX_Q) {
    // tab[w * Q + q] = 2^shift(w) C_q: 255 doublings per C_q, once per circuit, on the host
    std::vector<G1XYZZ> tx((size_t)CQ_TAB_W * Q);
    ThreadGroup th;
    const int nt = (int)std::min<long>(Q, 16);
    for (int t = 0; t < nt; t++)
      th.emplace_back([&, t] {
        for (long q = t; q < Q; q += nt) {
          G1XYZZ a = G1XYZZ::from_affine(cq[q]);
          tx[q] = a;
          for (int w = 1; w < CQ_TAB_W; w++) {
            for (int k = 0; k < msm_even_width(CQ_TAB_W, w - 1); k++) a = g1_dbl(a);
            tx[(size_t)w * Q + q] = a;
          }
        }
      });
    for (auto& x : th) x.join();
    std::vector<G1Affine> tab(tx.size());
    g1_batch_affine_host(tx.data(), (long)tx.size(), tab.data());
    p->cq_tab.alloc(sizeof(G1Affine) * tab.size());
    HIP_OK(hipMemcpy(p->cq_tab.p, tab.data(), sizeof(G1Affine) * tab.size(), hipMemcpyHostToDevice));
  }
  p->diag.resize(Q); p->yq.resize(Q);
  for (auto& b : p->diag) b.alloc(sizeof(Fr) * n);
  for (auto& b : p->yq) b.alloc(sizeof(Fr) * Q);
  p->slots.ensure(sizeof(MsmSlot) * (7 + 5 * Q + 1));
  p->prepared = true;
  API_END
}

void sonic_prover_free(sonic_prover_t* p) {
  if (!p) return;
  try { DeviceScope scope(p->device); delete p; } catch (const HipFail&) { delete p; }
}

// hscProve :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> m HscProof (Signature.hs:32-72) on its own, for the s(X,Y) of the handle's
// circuit (Constraints.hs:34-53) and ANY number m of (y_j, z_j) pairs -- inside prove() m is the number of linear constraints,
// the reference's own test of the sub-protocol (test/Test/Signature.hs:20-36) draws them freely.  u, v are hscProve's two `rnd`
// draws (:48, :60).  Not the hot path (prove() runs these MSMs interleaved with the rest of the proof): one group after the
// other on the handle's main stream, same kernels.  out: [S_j, s_j, W_j]_j, [s'_j, W'_j, Q_j]_j, Q_v, C, u, v.
size_t sonic_hsc_proof_size(int64_t m) { return (size_t)((2 + 4 * m) * 96 + (2 + 2 * m) * 32); }

int sonic_prover_hsc_prove(sonic_prover_t* p, int64_t m, const uint8_t* yzs, const uint8_t u[32], const uint8_t v[32], uint8_t* out) {
  API_BEGIN_ON(p ? p->device : -1)
  if (!p || m < 0 || (m > 0 && !yzs) || !u || !v || !out) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(p->mu);
  if (p->in_flight) { set_error("sonic_prover_hsc_prove: a submitted proof has not been collected yet"); return SONIC_ERR_INVALID_ARG; }
  const long n = p->n, Q = p->Q, d = srs_d(p->srs);
  for (long k = 0; k < 2 * m; k++)
    if (bytes_are_zero(yzs + 32 * k, 32)) { set_error("hscProve: y_j / z_j number %ld is zero: Laurent evaluation at 0 divides by zero", k); return SONIC_ERR_INEXACT_DIVISION; }
  if (bytes_are_zero(u, 32) || bytes_are_zero(v, 32)) { set_error("hscProve: u or v is zero"); return SONIC_ERR_INEXACT_DIVISION; }
  hipStream_t st = p->st;
  const sonic_srs* srs = p->srs;
  const Fr *wL = p->wL.as<Fr>(), *wR = p->wR.as<Fr>(), *wO = p->wO.as<Fr>();
  const long NS = 2 * m + 2;                                   // scalars: y_1..y_m, z_1..z_m, u, v
  const long K = 4 * m + 2;                                    // MSMs
  DevBuf S(sizeof(Fr) * NS), PR(sizeof(Fr) * 2 * NS), slots(sizeof(MsmSlot) * K), frout(sizeof(Fr) * (2 * m + 1)), flags(4);
  DevBuf pw(sizeof(Fr) * (3 * n + Q + 2)), su(sizeof(Fr) * (2 * n + Q + 1)), sy(sizeof(Fr) * (3 * n + 1));
  int* fl = flags.as<int>();
  HIP_OK(hipMemsetAsync(fl, 0, 4, st));
  {
    std::vector<uint8_t> h(32 * (size_t)NS);
    for (long j = 0; j < m; j++) { memcpy(&h[32 * j], yzs + 64 * j, 32); memcpy(&h[32 * (m + j)], yzs + 64 * j + 32, 32); }
    memcpy(&h[32 * (2 * m)], u, 32); memcpy(&h[32 * (2 * m + 1)], v, 32);
    HIP_OK(hipMemcpyAsync(S.p, h.data(), h.size(), hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
  }
  fr_to_mont_enqueue(st, S.as<Fr>(), NS, fl);
  fr_with_inverse_enqueue(st, S.as<Fr>(), (int)NS, PR.as<Fr>());
  const Fr* P0 = PR.as<Fr>();
  auto pY = [&](long j) { return P0 + 2 * j; };
  auto pZ = [&](long j) { return P0 + 2 * (m + j); };
  const Fr *pU = P0 + 2 * (2 * m), *pV = P0 + 2 * (2 * m + 1);
  MsmSlot* sl = slots.as<MsmSlot>();
  Fr* fo = frout.as<Fr>();
  Lane& lane = p->lane_at(0);
  const long s_lo = -n, s_len = 3 * n + 1, u_lo = -n, u_len = 2 * n + Q + 1;
  // slots: S_j = 3j, W_j = 3j + 1, W'_j = 3j + 2;  Q_j = 3m + j;  Q_v = 4m;  C = 4m + 1.   frout: s_j = j, s'_j = m + j
  for (long j = 0; j < m; j++) {                                                       // Signature.hs:40-45, 54
    poly_scale_powers_enqueue(st, nullptr, pw.as<Fr>(), 2 * n + Q + 1, -n, pY(j), pY(j) + 1);
    s_of_y_enqueue(st, wL, wR, wO, pw.as<Fr>(), n, Q, sy.as<Fr>());
    MsmJob jobs[3];
    jobs[0] = commit_job(st, srs, sy.as<Fr>(), s_lo, s_len, d, &sl[3 * j], fl);
    jobs[1] = open_job(st, srs, lane.sc[1], sy.as<Fr>(), s_lo, s_len, pZ(j), &fo[j], &sl[3 * j + 1], fl);
    jobs[2] = open_job(st, srs, lane.sc[2], sy.as<Fr>(), s_lo, s_len, pU, nullptr, &sl[3 * j + 2], fl);
    run_jobs(st, srs, lane.ws, jobs, 3);
  }
  poly_scale_powers_enqueue(st, nullptr, pw.as<Fr>(), 3 * n + 1, -n, pU, pU + 1);      // u^e, e in [-n, 2n]       :51
  s_of_u_enqueue(st, wL, wR, wO, pw.as<Fr>(), n, Q, su.as<Fr>(), p->tmp);
  {
    MsmJob jobs[MSM_MAX_JOBS];
    int k = 0;
    auto flush = [&] { run_jobs(st, srs, lane.ws, jobs, k); k = 0; };
    jobs[k++] = commit_job(st, srs, su.as<Fr>(), u_lo, u_len, d, &sl[4 * m + 1], fl);                               // C    :52
    for (long j = 0; j < m; j++) {                                                                                    // Q_j  :55
      if (k == MSM_MAX_JOBS) flush();
      jobs[k] = open_job(st, srs, lane.sc[k], su.as<Fr>(), u_lo, u_len, pY(j), &fo[m + j], &sl[3 * m + j], fl);
      k++;
    }
    if (k == MSM_MAX_JOBS) flush();
    jobs[k] = open_job(st, srs, lane.sc[k], su.as<Fr>(), u_lo, u_len, pV, &fo[2 * m], &sl[4 * m], fl);               // Q_v  :63
    k++;
    flush();
  }
  fr_from_mont_enqueue(st, fo, 2 * m + 1);
  std::vector<MsmSlot> hs((size_t)K);
  std::vector<uint8_t> hfr(32 * (size_t)(2 * m + 1));
  HIP_OK(hipMemcpyAsync(hs.data(), sl, sizeof(MsmSlot) * K, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(hfr.data(), fo, hfr.size(), hipMemcpyDeviceToHost, st));
  int hflags = read_flags(st, flags);
  if (hflags) return flags_to_status(hflags, "hscProve");
  std::vector<uint8_t> pts(96 * (size_t)K);
  {
    std::vector<G1XYZZ> sums((size_t)K);
    bool folded = true;
    for (long i = 0; i < K; i++) folded = folded && (hs[i].W == 1 || hs[i].pad1 != 0);
    if (folded) {
      for (long i = 0; i < K; i++) sums[i] = msm_finish_host(hs[i]);
    } else {
      ThreadGroup th;
      const int nt = (int)std::min<long>(K, 16);
      for (int w = 0; w < nt; w++) th.emplace_back([&, w] { for (long i = w; i < K; i += nt) sums[i] = msm_finish_host(hs[i]); });
      for (auto& x : th) x.join();
    }
    g1_canonical_bytes_host_batch(sums.data(), (int)K, pts.data());
  }
  uint8_t* o = out;
  auto putG = [&](long i) { memcpy(o, &pts[96 * (size_t)i], 96); o += 96; };
  auto putF = [&](const uint8_t* b) { memcpy(o, b, 32); o += 32; };
  for (long j = 0; j < m; j++) { putG(3 * j); putF(&hfr[32 * j]); putG(3 * j + 1); }                 // hscS
  for (long j = 0; j < m; j++) { putF(&hfr[32 * (m + j)]); putG(3 * j + 2); putG(3 * m + j); }       // hscW
  putG(4 * m); putG(4 * m + 1); putF(u); putF(v);                                                   // Qv, C, u, v
  API_END
}

}  // extern "C"
