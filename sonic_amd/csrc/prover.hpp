// What the translation units of the prover share (prove.hip: one proof on one handle; prove_multi.hip: the one-shot call and the calls that
// take several handles; poly_api.hip: commitPoly / openPoly / hscProve on caller-supplied polynomials, NTT and dense product): the handle,
// its lanes, the opening scratch, and the helpers that turn a polynomial into the MSM that commits to it or opens it.  Internal: included
// by those three files only (hence the using-directive below).
#pragma once
#include <string.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <memory>
#include <thread>
#include <numeric>
#include <string>
#include <vector>
#include "internal.hpp"
#include "poly.hpp"
#include "fs.hpp"
#include "share_plan.hpp"


namespace sonic {


enum { FLAG_BAD_ENCODING = 1, FLAG_SRS_INDEX = 2 };
// phases of a proof = the points at which the reference's prover draws (Protocol.hs:58,66,76,84-85; Signature.hs:48,60): what can be
// computed once the draws up to there are known
enum { PH_R = 1,      // R                                   (blinders)
       PH_T = 2,      // T                                   (+ y)
       PH_OPEN = 3,   // a, W_a, b, W_b, W_t, s              (+ z)
       PH_HSCS = 4,   // S_j, s_j, W_j                       (+ y_j, z_j)
       PH_HSCW = 5,   // C, s'_j, W'_j, Q_j                  (+ u)
       PH_QV = 6,     // Q_v                                 (+ v)
       PH_ALL = 0x7e };
// Buckets per running-sum segment inside prove().  More buckets per segment = fewer small scalar multiplications (less work),
// fewer = shorter dependent chains.  Batched groups hide their chains under other groups' accumulation, so they take the
// work-optimal end; the group that finishes last has nothing left to hide under and takes a short chain.  Measured
// (ms per proof, batched / last): n = 2^18 (2^19 buckets): 16/8 45.9, 32/8 44.3, 64/8 43.2, 64/16 42.6, 64/32 43.7, 128/16 44.6;
// n = 2^16 (2^16 buckets): 4/4 15.7, 8/4 15.1, 16/4 14.95, 16/8 15.1; n = 2^14: 4/4 7.2, 8/4 6.7, 16/4 6.6.
// window tables of the per-circuit commitments C_q (sonic_prover_prepare): 29 windows of 9 / 8 bits over 256 shared buckets
constexpr int CQ_TAB_W = 29, CQ_TAB_C = 9;
constexpr long CQ_TAB_MAX_Q = 1L << 16;
struct Scratch {
  DevBuf D, q, scan, fz_discard;
  void reserve(long len) { D.ensure(sizeof(Fr) * (len + 1)); q.ensure(sizeof(Fr) * (len + 1)); fz_discard.ensure(sizeof(Fr)); }
};

// (prove.hip)
MsmJob commit_job(hipStream_t st, const sonic_srs* srs, const Fr* poly, long lo, long len, long maxm, MsmSlot* slot, int* d_flags);
void run_jobs(hipStream_t st, const sonic_srs* srs, MsmWorkspace& ws, const MsmJob* jobs, int k, bool last = false, bool exposed = true);
MsmJob open_job(hipStream_t st, const sonic_srs* srs, Scratch& sc, const Fr* poly, long lo, long len,
                       const Fr* zpair, Fr* d_fz, MsmSlot* slot, int* d_flags);
MsmJob open_job_at_zero(hipStream_t st, const sonic_srs* srs, const Fr* poly, long len, Fr* d_fz, MsmSlot* slot, int* d_flags);
bool bytes_are_zero(const uint8_t* p, size_t n);

}  // namespace sonic

using namespace sonic;

// One MSM "lane": its own stream, bucket workspace and opening scratch.  The MSMs of a proof that depend on the same
// polynomial form a group (R, W_a, W_b | T, W_t | S_j, W_j, W'_j | C, Q_j.., Q_v) that runs on one lane as ONE batched
// kernel chain (msm_enqueue_batch); different groups run on different lanes so that one group's sort and reduction
// phases run under another's accumulation.
struct Lane {
  hipStream_t st = nullptr;
  hipEvent_t done = nullptr;
  hipEvent_t prep = nullptr;     // fused proofs: the lane's openings (evaluation, quotient) are queued; the proof's ONE chain waits for it
  MsmWorkspace ws;
  Scratch sc[MSM_MAX_JOBS];       // one per opening of the group in flight (grown on first use)
  MsmJob jobs[MSM_MAX_JOBS];
  int njobs = 0;
};
constexpr int N_LANES = 6;

struct sonic_prover {
  const sonic_srs* srs = nullptr;
  int device = 0;                            // the SRS's GPU: the handle's streams, buffers and every call on it live there
  long n = 0, Q = 0;
  hipStream_t st = nullptr;
  hipStream_t ts = nullptr;                  // the t(X,y) product (NTT) runs beside the hscProve polynomials
  bool have_assignment = false;
  DevBuf wL, wR, wO, cs, aL, aR, aO;        // Montgomery, resident across proofs
  Lane lanes[N_LANES];
  // lanes in use: all six by default.  SONIC_FUSED_LANES=k (small-proof handles): k lanes of their own; =0: NO lane of its own
  // (few_streams) -- three streams per handle: main, transform, chain; the groups' openings ride on the streams that are waiting anyway
  // (r(X,1)'s on the transform's stream ahead of the product, the s(X,y_j) groups' and s(u,Y)'s on the main stream behind the polynomials,
  // t(X,y)'s behind the product) through two lanes that only borrow those streams.  Why the knob exists: the runtime multiplexes a process's
  // streams onto 8 hardware queues, a new stream getting the least-used one, so WHICH of two streamed handles' twenty streams share a queue
  // is luck -- one handle's openings behind the other handle's accumulation cost 10 % -- and more hardware queues are worse
  // (GPU_MAX_HW_QUEUES = 12 .. 32: +1 ms on a sequential small proof, profiles/r06_ab_queues.txt).  Measured, 64 proofs at n = 2^16 on one
  // box, four fresh pairs of handles each (profiles/r06_batch_mode.txt): six lanes 104-107 proofs/s; three streams 99.2-100.5, every time;
  // one lane + the borrowed streams (four streams) 98-111 depending on the order the handles were made in.  Three streams lose the
  // side-by-side openings (n = 2^14 streamed 3.8 against 3.45 ms, n = 2^10 1.5 against 1.2), so six lanes stay the default.
  int n_lanes = N_LANES;
  bool few_streams = false;
  Lane main_lane, ts_lane;                   // st = the handle's main / transform stream (not owned)
  // fused proofs (below): the proof's ONE chain runs on a stream of its own.  (Stream priorities -- the chain lowest, everything that builds
  // polynomials and openings highest, so that the next streamed proof's preparation would get wave slots beside a running accumulation --
  // were measured and made things WORSE on this runtime: n = 2^16 streamed 10.6 against 9.75 ms, n = 2^14 4.1 against 3.55,
  // profiles/r06_ab_small.txt; SONIC_PROVE_PRIORITIES=1 still asks for them.)
  Lane chain[2];
  bool small_plan = false;                   // decided when the handle is made: the SRS plans 2^17 buckets or fewer for this n
  int next_lane = 0;
  const NttTables* ntt = nullptr;            // the device's shared tables for 2^log2m points (device_ntt_tables)
  DevBuf S, PAIRS, r1, sy0, su, pw, kpow, fa, fb, slots, frout, flags, tmp;
  std::vector<DevBuf> syj;
  // sonic_prover_prepare: Commit(P_q) per constraint row (affine, Montgomery) and per-j scalar buffers
  // pinned host staging (fixed addresses: the whole enqueue of a proof can be captured once and replayed as a hipGraph)
  uint8_t* h_tr = nullptr;
  Fr* h_pairs = nullptr;         // {v, v^-1} of the evaluation points, computed on the host (prove_enqueue)
  MsmSlot* h_slots = nullptr;
  uint8_t* h_fr = nullptr;
  int* h_flags = nullptr;                  // [0] the proof's flags, [1] those of a circuit uploaded inside the proof (pend_circuit)
  // one-shot calls into a parked shell (sonic_prove): the circuit of THIS call, still in the caller's host buffers.  prove_enqueue
  // uploads it after it has queued the group of MSMs that needs the assignment only (R, W_a, W_b), so the 2 Q n + Q weights cross
  // PCIe under those kernels instead of in front of the proof
  const uint8_t* pend_circuit[4] = {nullptr, nullptr, nullptr, nullptr};
  // the same for the ASSIGNMENT of this call (round 6: sonic_prove_batch with per-proof assignments, sonic_prove): uploaded at the head of
  // the proof's own queue instead of by a sonic_prover_set_assignment that waits for the device before the proof may even be queued -- beside
  // another handle's accumulation that wait was ~1 ms per proof (config5: 99 against 113 proofs/s, profiles/r06_bench.json)
  const uint8_t* pend_asg[3] = {nullptr, nullptr, nullptr};
  // runs of equal coefficients in the S_j of a handle that is not prepared (poly.hip, k_run_tiles): per j the masked copy of s(X, y_j),
  // the tile records and the (scalar, running-sum point) slots of the small MSM that stands for the runs; its sum lands in slot
  // (7 + 4Q) + j, where a prepared handle keeps sum_q y_j^{n+q} C_q, and the host adds it the same way
  // The small MSM runs on the handle's MAIN stream, which has built all polynomials by then and only waits for the lanes: behind the
  // batch of the lane that reads the masked copy its ~13 short launches and the chained sums of its per-window buckets added 0.6-1.3 ms
  // to a proof's latency; a stream of its own (two more streams per handle than the 8 hardware queues the runtime is given) cost
  // 1-5 ms per streamed proof.
  struct RunBufs {
    DevBuf masked, val, uniform, scal, pts;
    hipEvent_t masked_ev = nullptr;
  };
  MsmWorkspace runs_ws;
  std::vector<RunBufs> runs;
  bool runs_on = false;
  // does this circuit HAVE runs?  Sampled on the host from the weights as they are handed over (circuit_runs_hint): a circuit without
  // repeated rows would pay the masked copy, the tile scan and a small MSM per S_j and get nothing back (measured with
  // uniformly random weights at n = 2^18: profiles/r06_runs_dense_ab.txt).  A hint only: the path is exact for any input.
  bool circuit_has_runs = true;
  // C = commitPoly(s(u, Y)) through the SRS's symmetric sums (srs.hip, srs_build_sym): s(u, Y) has the same coefficient at Y^i and Y^-i
  // (i <= n), so n terms over A[i] + A[-i] and a Q-term MSM for Y^{n+1} .. Y^{n+Q} stand for its 2n + Q + 1 terms; the Q-term sum lands
  // in slot 7 + 5Q and the host adds it
  bool sym_on = false;
  // Small proofs (round 6): ALL the MSMs of a proof as ONE batched kernel chain.  With 2^16 shared buckets and fewer (c <= 18: d < 2^20,
  // n <= 2^16) a group of two or three MSMs is 2048-3072 one-thread-per-bucket waves -- one round of the chip's 2048 wave slots, half
  // empty on its second -- and a proof is five such chains whose sorts, heavy-bucket launches and reductions (each ~15 launches of
  // 5-20 us, plus ~1.4 ms of latency-bound running sums) queue behind each other's accumulation for wave slots: n = 2^14 measured 5.5 ms
  // per proof for 1.8 ms of additions at the full-chip rate, n = 2^16 11.1 ms for 7.2 (profiles/r06_small_proofs.txt).  Fused, the lanes
  // only prepare the openings (evaluation, prefix sums, quotient: they still run side by side); their jobs are collected here and run as
  // one chain of up to MSM_MAX_JOBS jobs on the t lane: one sort, ONE accumulation launch that keeps every wave slot filled until its
  // tail, one butterfly over all bucket sets.  Larger plans (2^19 buckets per set) fill the chip per group and keep the lanes
  // (packing their groups was measured slower in round 3).  SONIC_PROVE_FUSED=0 / 1: never / whenever the plan batches.
  bool fused = false;
  std::vector<MsmJob> fused_jobs;
  std::vector<std::unique_ptr<Scratch>> fused_sc;      // one per opening of the proof: a quotient lives until the chain has read it
  size_t fused_sc_next = 0;
  Scratch& fused_scratch() {
    if (fused_sc_next == fused_sc.size()) fused_sc.emplace_back(new Scratch());
    return *fused_sc[fused_sc_next++];
  }
  // SONIC_PROVE_GRAPH=1 (read when the handle is made): capture the enqueue of the second proof and replay it.  Off by default: in round 3
  // the replay of a ~220-node, 7-stream proof was slower than the direct launches (n = 2^10: 9.0 vs 4.9 ms per proof, n = 2^14: 9.6 vs 6.5);
  // with one chain per proof (round 6) it is 4-8 % faster streamed from one host thread (n = 2^14: 3.18 vs 3.47 ms), 5-20 % slower one at a
  // time, and no faster through sonic_prove_batch at BASELINE configs[4] (profiles/r06_ab_graph.txt).
  bool use_graph = false;
  hipGraphExec_t graph = nullptr;
  bool graph_tried = false;
  long proofs_done = 0;
  bool in_flight = false;                    // between sonic_prover_submit and sonic_prover_collect
  // Which steps of the proof an enqueue runs (bit = phase; PH_ALL normally).  The Fiat-Shamir mode (sonic_prover_prove_fs) proves in
  // six passes, each running exactly the MSMs whose challenges have become known: results of earlier passes stay in `slots` / `frout`.
  uint32_t phases = 0x7e;
  DevBuf frstd;                              // frout in standard form (frout itself stays Montgomery across passes)
  std::chrono::steady_clock::time_point t_begin, t_enq;
  bool prepared = false;
  DevBuf cq;
  // window tables of the C_q (CQ_TAB_W x Q points, table w = 2^shift(w) C_q): their Q-term MSM then shares one bucket set and
  // leaves ONE window sum like every other MSM of a proof, instead of 64 that the host folds with 255 doublings (126 us each)
  DevBuf cq_tab;
  std::vector<DevBuf> diag, yq;
  hipEvent_t ev_r1 = nullptr, ev_sy0 = nullptr, ev_t = nullptr, ev_su = nullptr;
  std::vector<hipEvent_t> ev_syj;
  int log2m = 0;
  std::mutex mu;
  // ONE proof over several GPUs (sonic_prover_set_share, share_plan.hpp): this handle runs rank share_rank's pieces of the proof's
  // MSMs and reports un-normalised partial sums (sonic_prover_collect_share); share_world <= 1: the whole proof
  int share_rank = 0, share_world = 0;
  SharePlan share;
  bool share_planned_prepared = false, share_planned = false;
  int32_t share_plan_tag = 0;                // hash of the plan's inputs (share header: ranks must have planned alike)
  std::vector<uint8_t> slot_ran;             // per slot (7 + 5Q): the last enqueue queued an MSM for it
  std::vector<uint8_t> fr_valid;             // per evaluation (3 + 2Q): the last enqueue computed it
  uint8_t witness_digest[32] = {0};          // SHA-256 of the assignment (Fiat-Shamir blinders, fs.hpp), made on first use
  bool have_witness_digest = false;
  // Lane N_LANES-1 carries the t(X,y) group (the largest, ready last); the other groups alternate over the rest, which
  // balances the point additions per lane (Q = 2: 55M / 51M / 48M) while one lane's sort and reduction phases run under
  // another lane's accumulation.  Streams beyond the 4 hardware queues would serialise behind each other.
  Lane& lane_at(int i) { return few_streams ? ((i & 1) ? ts_lane : main_lane) : lanes[i % n_lanes]; }      // (prepare, hscProve: any lane)
  Lane& pick(hipEvent_t ready) {
    if (few_streams) return main_lane;
    Lane& l = lanes[next_lane];
    next_lane = (next_lane + 1) % (n_lanes > 1 ? n_lanes - 1 : 1);
    (void)hipStreamWaitEvent(l.st, ready, 0);
    return l;
  }
  Lane& t_lane(hipEvent_t ready) {
    // (few_streams: the transform's own stream, which has waited for `ready` when it took the polynomials in -- a second wait for the same
    // event, or one for an event of the stream itself, becomes a duplicate edge when the enqueue is captured as a hipGraph, and the
    // runtime's capture code crashed on it)
    if (few_streams) return ts_lane;
    Lane& l = lanes[n_lanes - 1];
    (void)hipStreamWaitEvent(l.st, ready, 0);
    return l;
  }
  ~sonic_prover() {
    for (auto& l : lanes) { if (l.st) (void)hipStreamDestroy(l.st); if (l.done) (void)hipEventDestroy(l.done); if (l.prep) (void)hipEventDestroy(l.prep); }
    for (auto& l : chain) { if (l.st) (void)hipStreamDestroy(l.st); if (l.done) (void)hipEventDestroy(l.done); if (l.prep) (void)hipEventDestroy(l.prep); }
    for (Lane* l : {&main_lane, &ts_lane}) { if (l->done) (void)hipEventDestroy(l->done); if (l->prep) (void)hipEventDestroy(l->prep); }
    for (auto& r : runs) if (r.masked_ev) (void)hipEventDestroy(r.masked_ev);
    for (hipEvent_t e : {ev_r1, ev_sy0, ev_t, ev_su}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_syj) if (e) (void)hipEventDestroy(e);
    if (st) (void)hipStreamDestroy(st);
    if (ts) (void)hipStreamDestroy(ts);
    if (graph) (void)hipGraphExecDestroy(graph);
    for (void* h : {(void*)h_tr, (void*)h_pairs, (void*)h_slots, (void*)h_fr, (void*)h_flags}) if (h) (void)hipHostFree(h);
  }
};

#define API_BEGIN_ON(dev) try { ::sonic::DeviceScope _scope(dev);
#define API_BEGIN API_BEGIN_ON(-1)
#define API_END                                                        \
  } catch (const HipFail& f) { return f.code; }                        \
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; } \
  return SONIC_OK;

// (prove.hip)
int upload_fr_mont(hipStream_t st, DevBuf& dst, const uint8_t* src, long count, int* d_flags);
int read_flags(hipStream_t st, DevBuf& flags);
int flags_to_status(int f, const char* who);
// (defined inside prove.hip's extern "C" block, not exported)
extern "C" int prove_with_assignment(sonic_prover_t* p, const uint8_t* aL, const uint8_t* aR, const uint8_t* aO, const uint8_t* transcript, uint8_t* out_proof);
bool circuit_runs_hint(const uint8_t* wL, const uint8_t* wR, long n, long Q);

