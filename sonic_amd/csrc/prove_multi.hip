// The calls that wrap whole proofs around the handle of prove.hip: the reference's own call shape with everything handed over per call
// (sonic_prove: src/Sonic/Protocol.hs:47-52; parked shells), many statements over several SRS replicas (sonic_prove_many), ONE proof shared
// by several handles (sonic_prove_shared) and a batch of proofs over several handles (sonic_prove_batch: `mapM prove`).
#include "prover.hpp"

// prove :: SRS -> Assignment Fr -> ArithCircuit Fr -> m (Proof, RndOracle) with the reference's own shape: everything handed over per
// call (Protocol.hs:47-52).  A handle costs streams, events, ~20 workspace allocations that grow on the first proof and the twiddle
// tables -- tens of milliseconds against a 32-ms proof -- so the device parks the shell of a finished one-shot call and the next call
// with the same SRS handle and (n, Q) only uploads its circuit and assignment into it (round 5; bench.py `one_shot`).  Several host
// threads inside sonic_prove on one GPU each take a parked shell or make one; at most ONE_SHOT_SHELLS stay parked per device.
namespace {
constexpr size_t ONE_SHOT_SHELLS = 4;
}
namespace sonic {
void drop_one_shot_of(const sonic_srs* s) {
  DeviceCtx& c = current_ctx();
  std::vector<OneShotShell*> gone;
  {
    std::lock_guard<std::mutex> g(c.one_shot_mu);
    for (size_t i = 0; i < c.one_shot.size();) {
      OneShotShell* sh = static_cast<OneShotShell*>(c.one_shot[i]);
      if (sh->srs == s) { gone.push_back(sh); c.one_shot.erase(c.one_shot.begin() + (long)i); } else i++;
    }
  }
  for (OneShotShell* sh : gone) { delete sh->p; delete sh; }
}
}
extern "C" {
// frees the parked one-shot shells of every device this process has used (or of one device: device >= 0).  A parked shell holds all the
// workspaces of a proof of its shape -- several GB at n = 2^20 -- until another shape evicts it or its SRS is freed; a host that has
// finished a burst of sonic_prove / sonic_prove_many calls gives the memory back with this.  Returns the number of shells freed.
int sonic_one_shot_trim(int device) {
  int freed = 0;
  try {
    int ndev = 0;
    if (sonic_device_count(&ndev) != SONIC_OK) return 0;
    for (int d = 0; d < ndev; d++) {
      if (device >= 0 && d != device) continue;
      DeviceScope scope(d);
      DeviceCtx& c = scope.ctx();
      std::vector<void*> gone;
      { std::lock_guard<std::mutex> g(c.one_shot_mu); gone.swap(c.one_shot); }
      for (void* v : gone) { OneShotShell* sh = static_cast<OneShotShell*>(v); delete sh->p; delete sh; freed++; }
    }
  } catch (const HipFail&) {}
  return freed;
}

int sonic_prove(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
                const uint8_t* cs, const uint8_t* aL, const uint8_t* aR, const uint8_t* aO, const uint8_t* transcript,
                uint8_t* out_proof) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || n < 1 || Q < 1 || !wL || !wR || !wO || !cs || !aL || !aR || !aO || !transcript || !out_proof) { set_error("sonic_prove: bad argument (need n >= 1, Q >= 1)"); return SONIC_ERR_INVALID_ARG; }
  DeviceCtx& ctx = current_ctx();
  OneShotShell* sh = nullptr;
  {
    std::lock_guard<std::mutex> g(ctx.one_shot_mu);
    for (size_t i = ctx.one_shot.size(); i-- > 0;) {                 // newest first
      OneShotShell* c = static_cast<OneShotShell*>(ctx.one_shot[i]);
      if (c->srs == srs && c->p->n == n && c->p->Q == Q) { sh = c; ctx.one_shot.erase(ctx.one_shot.begin() + (long)i); break; }
    }
  }
  int rc = SONIC_OK;
  if (sh) {   // uploaded inside the proof
    sh->p->pend_circuit[1] = wR; sh->p->pend_circuit[2] = wO; sh->p->pend_circuit[3] = cs; sh->p->pend_circuit[0] = wL;
    sh->p->circuit_has_runs = circuit_runs_hint(wL, wR, n, Q);
  }
  else {
    sonic_prover_t* p = nullptr;
    rc = sonic_prover_new(srs, n, Q, wL, wR, wO, cs, &p);
    if (rc) return rc;
    sh = new OneShotShell{srs, p};
  }
  if (!rc) rc = prove_with_assignment(sh->p, aL, aR, aO, transcript, out_proof);
  sh->p->pend_circuit[0] = nullptr;           // (a call that failed before its upload: the caller's buffers end with the call)
  // park the shell (also after a failed call: the next one loads its own circuit and assignment); the oldest parked shell makes room
  OneShotShell* evict = nullptr;
  {
    std::lock_guard<std::mutex> g(ctx.one_shot_mu);
    if (ctx.one_shot.size() >= ONE_SHOT_SHELLS) { evict = static_cast<OneShotShell*>(ctx.one_shot.front()); ctx.one_shot.erase(ctx.one_shot.begin()); }
    ctx.one_shot.push_back(sh);
  }
  if (evict) { delete evict->p; delete evict; }
  return rc;
  API_END
}

// mapM (\(assignment, circuit) -> prove srs assignment circuit) over K INDEPENDENT statements of one shape (n, Q), spread over the SRS
// replicas -- one per GPU -- with two host threads per replica (statement i on thread i mod 2 n_srs; two one-shot calls in flight per
// GPU stream it: one call's upload and host tail under the other's kernels).  BASELINE's "batch of 64 independent proofs streamed over 8
// GPUs" with every proof its own circuit and witness; no collective.  Returns the first non-zero status in list order; all are attempted.
int sonic_prove_many(const sonic_srs_t* const* srs, int n_srs, int64_t n, int64_t Q, const sonic_statement_t* statements, int64_t K,
                     uint8_t* out_proofs, int* out_status) {
  if (!srs || n_srs < 1 || n_srs > 512 || n < 1 || Q < 1 || K < 0 || (K > 0 && (!statements || !out_proofs))) return SONIC_ERR_INVALID_ARG;
  for (int i = 0; i < n_srs; i++) if (!srs[i]) return SONIC_ERR_INVALID_ARG;
  const size_t psz = sonic_proof_size(Q);
  const int T = 2 * n_srs;
  std::vector<int> status((size_t)K, SONIC_OK);
  std::vector<std::string> errs((size_t)T);
  auto body = [&](int t) {
    const sonic_srs_t* s = srs[t % n_srs];
    for (int64_t i = t; i < K; i += T) {
      int rc = SONIC_ERR_HIP;
      try {
        const sonic_statement_t& st = statements[i];
        rc = sonic_prove(s, n, Q, st.wL, st.wR, st.wO, st.cs, st.aL, st.aR, st.aO, st.transcript, out_proofs + psz * (size_t)i);
        if (rc && errs[(size_t)t].empty()) { char b[512]; sonic_last_error(b, sizeof b); errs[(size_t)t] = b; }
      } catch (...) { rc = SONIC_ERR_HIP; }                            // (nothing may leave a thread's body: std::terminate)
      status[(size_t)i] = rc;
    }
  };
  {
    ThreadGroup th;
    for (int t = 1; t < T && t < K; t++) th.emplace_back(body, t);
    body(0);
    for (auto& x : th) x.join();
  }
  if (out_status) for (int64_t i = 0; i < K; i++) out_status[i] = status[(size_t)i];
  for (int64_t i = 0; i < K; i++)
    if (status[(size_t)i]) {
      const int t = (int)(i % T);
      set_error("sonic_prove_many, statement %ld (device %d): %s", (long)i, srs_device(srs[t % n_srs]), errs[(size_t)t].c_str());
      return status[(size_t)i];
    }
  return SONIC_OK;
}

// ---- N GPUs from ONE host process: one proof shared by several handles, a batch of proofs over several handles ------------------
// (include/sonic_hip.h).  One host thread per handle; a handle carries its device, so the threads need no set-up of their own.
int sonic_prover_device(const sonic_prover_t* p) { return p ? p->device : -1; }

int sonic_prove_shared(sonic_prover_t* const* provers, int world, const uint8_t* transcript, uint8_t* out_proof) {
  if (!provers || world < 1 || world > 1024 || !transcript || !out_proof) return SONIC_ERR_INVALID_ARG;
  for (int r = 0; r < world; r++) {
    if (!provers[r]) return SONIC_ERR_INVALID_ARG;
    if (provers[r]->n != provers[0]->n || provers[r]->Q != provers[0]->Q) { set_error("sonic_prove_shared: handle %d proves another circuit shape (n, Q) than handle 0", r); return SONIC_ERR_INVALID_ARG; }
    for (int q = 0; q < r; q++) if (provers[q] == provers[r]) { set_error("sonic_prove_shared: handle %d appears twice (one handle runs one share at a time)", r); return SONIC_ERR_INVALID_ARG; }
  }
  if (world == 1) {
    // one handle: the whole proof (a handle left in share mode by an earlier call goes back first)
    if (provers[0]->share_world > 1) { int rc = sonic_prover_set_share(provers[0], 0, 1); if (rc) return rc; }
    return sonic_prover_prove(provers[0], transcript, out_proof);
  }
  const long Q = provers[0]->Q;
  const size_t ssz = sonic_proof_share_size(Q);
  std::vector<uint8_t> shares(ssz * (size_t)world);
  std::vector<int> rcs((size_t)world, SONIC_OK);
  std::vector<std::string> errs((size_t)world);
  auto body = [&](int r) {
    sonic_prover_t* p = provers[r];
    int rc = SONIC_OK;
    try {
      bool placed;
      { std::lock_guard<std::mutex> g(p->mu); placed = p->share_world == world && p->share_rank == r; }
      if (!placed) rc = sonic_prover_set_share(p, r, world);          // (clears the slots of pieces the handle no longer runs; once per change)
      if (!rc) rc = sonic_prover_prove_share(p, transcript, &shares[ssz * (size_t)r]);
      if (rc) { char b[512]; sonic_last_error(b, sizeof b); errs[(size_t)r] = b; }
    } catch (...) { rc = SONIC_ERR_HIP; }                            // (nothing may leave a thread's body: std::terminate)
    rcs[(size_t)r] = rc;
  };
  {
    ThreadGroup th;
    for (int r = 1; r < world; r++) th.emplace_back(body, r);
    body(0);                                                       // the calling thread is rank 0's
    for (auto& t : th) t.join();
  }
  for (int r = 0; r < world; r++)
    if (rcs[(size_t)r]) { set_error("sonic_prove_shared, rank %d (device %d): %s", r, provers[r]->device, errs[(size_t)r].c_str()); return rcs[(size_t)r]; }
  return sonic_proof_from_shares(Q, world, shares.data(), transcript, out_proof);
}

int sonic_prove_batch(sonic_prover_t* const* provers, int n_provers, int64_t K, const uint8_t* aL, const uint8_t* aR, const uint8_t* aO,
                      const uint8_t* transcripts, uint8_t* out_proofs, int* out_status) {
  if (!provers || n_provers < 1 || n_provers > 1024 || K < 0 || (K > 0 && (!transcripts || !out_proofs))) return SONIC_ERR_INVALID_ARG;
  const bool per_proof = aL || aR || aO;
  if (per_proof && !(aL && aR && aO)) { set_error("sonic_prove_batch: aL, aR, aO must be given together (or all NULL: the handles' resident assignments)"); return SONIC_ERR_INVALID_ARG; }
  for (int i = 0; i < n_provers; i++) {
    if (!provers[i]) return SONIC_ERR_INVALID_ARG;
    if (provers[i]->n != provers[0]->n || provers[i]->Q != provers[0]->Q) { set_error("sonic_prove_batch: handle %d proves another circuit shape (n, Q) than handle 0", i); return SONIC_ERR_INVALID_ARG; }
    if (provers[i]->share_world > 1) { set_error("sonic_prove_batch: handle %d runs one rank's share of a proof (sonic_prover_set_share)", i); return SONIC_ERR_INVALID_ARG; }
    for (int q = 0; q < i; q++) if (provers[q] == provers[i]) { set_error("sonic_prove_batch: handle %d appears twice", i); return SONIC_ERR_INVALID_ARG; }
  }
  const long n = provers[0]->n, Q = provers[0]->Q;
  const size_t psz = sonic_proof_size(Q), tsz = 32 * (size_t)(8 + 2 * Q), asz = 32 * (size_t)n;
  std::vector<int> status((size_t)K, SONIC_OK);
  std::vector<std::string> errs((size_t)n_provers);
  std::vector<int64_t> first_bad((size_t)n_provers, -1);
  auto body = [&](int h) {
    for (int64_t i = h; i < K; i += n_provers) {
      int rc = SONIC_OK;
      try {
        if (per_proof) rc = prove_with_assignment(provers[h], aL + asz * (size_t)i, aR + asz * (size_t)i, aO + asz * (size_t)i, transcripts + tsz * (size_t)i, out_proofs + psz * (size_t)i);
        else rc = sonic_prover_prove(provers[h], transcripts + tsz * (size_t)i, out_proofs + psz * (size_t)i);
        if (rc && first_bad[(size_t)h] < 0) { first_bad[(size_t)h] = i; char b[512]; sonic_last_error(b, sizeof b); errs[(size_t)h] = b; }
      } catch (...) { rc = SONIC_ERR_HIP; }                          // (nothing may leave a thread's body: std::terminate)
      status[(size_t)i] = rc;
    }
  };
  {
    ThreadGroup th;
    for (int h = 1; h < n_provers && h < K; h++) th.emplace_back(body, h);
    body(0);
    for (auto& t : th) t.join();
  }
  if (out_status) for (int64_t i = 0; i < K; i++) out_status[i] = status[(size_t)i];
  for (int64_t i = 0; i < K; i++)
    if (status[(size_t)i]) {
      const int h = (int)(i % n_provers);
      set_error("sonic_prove_batch, proof %ld (handle %d, device %d): %s", (long)i, h, provers[h]->device, errs[(size_t)h].c_str());
      return status[(size_t)i];
    }
  return SONIC_OK;
}

}  // extern "C"
