// Internal interface of the G1 multi-scalar multiplication (msm.hip).
#pragma once
#include "common.hpp"
#include "g1.hpp"
#include "g1_host.hpp"

namespace sonic {

// Window tables: W windows over the 255 bits a folded scalar and its recoding carries can occupy, with widths that differ by at
// most one bit (the first 255 % W windows are the wide ones).  With W = 13: eight 20-bit and five 19-bit windows.  A uniform
// width c would leave the top window 254 - (W-1) c bits: its 2^14 buckets then hold N / 2^14 extra terms each, the walks over
// them are 3x longer than the rest and set the duration of the whole accumulation (average residency 1.5 of 2 waves per SIMD).
// `bits`: what the windows cover -- 255 for a whole (folded) scalar, ENDO_BITS = 130 for the halves of an endomorphism split (endo.hpp)
HD int msm_even_width(int W, int w, int bits = 255) { return bits / W + (w < bits % W ? 1 : 0); }
HD int msm_even_shift(int W, int w, int bits = 255) { const int base = bits / W, extra = bits % W; return w * base + (w < extra ? w : extra); }

struct MsmPlan {
  int c;        // window bits
  int W;        // windows: ceil(255 / c) (scalars are first folded into [0, (r-1)/2])
  int Wb;       // bucket sets: W (one per window), or 1 when the windows share buckets over precomputed tables
  long table_stride;   // points per window table (0: no tables, every window reads the same points); with tables the windows
                       // have the even widths above and c is the widest
  int NB;       // buckets per set: digit magnitudes 1 .. 2^(c-1)
  int K;        // buckets per running-sum segment
  int nseg;     // segments per window
  uint32_t heavy_threshold;
  // threads per workgroup of the bucket accumulation.  256 inside prove() (several chains run side by side; measured in round 3:
  // 34.7 ms per proof against 35.8-36.2 with 64 or 128); 64 for the stand-alone MSM lanes, where one-wave workgroups refill a freed
  // wave slot without waiting for three more (k_bucket_accum 2.31-2.40 against 2.42-2.45 ms, MSM 2^20 streamed 3.22-3.24 against
  // 3.33-3.59 ms).
  int accum_block = 256;
  // reduce the shared bucket set by the bit-sum butterfly (log-depth, least work, ~4x the memory traffic) instead of running sums over
  // K-bucket segments (one pass over the buckets, a long dependent chain per segment).  The tree wins wherever the reduction is
  // exposed -- a stand-alone MSM, the last group of a proof, a rank's few pieces of a shared proof -- the segments where it hides under
  // other groups' accumulation (profiles/r04_bucket_tree.txt).
  bool tree = true;
  // endomorphism plan (endo.hpp): the tables hold W windows over 130 bits; every job runs as TWO device jobs over the same points,
  // with the halves s1, s2 of its scalars, and the host adds phi(second sum) to the first.  Tree reduction only.
  bool endo = false;
  int bits = 255;
  // fold scalars s > (r-1)/2 into r - s on the negated point: valid iff r P = O.  Every SRS element is in the r-torsion;
  // caller-supplied points of sonic_msm_g1 only have to be on the curve (E(Fq) has cofactor points, e.g. (0, 2) of order 3),
  // and s P then means the literal multiple the reference's `mul` computes, so that entry point does not fold.
  bool fold;
};
MsmPlan msm_plan(long n, bool fold = true);
MsmPlan msm_plan_tables(long n, int c, int W, long table_stride, bool endo = false);
// Trade latency for work in the bucket running sums: K buckets per segment.  A standalone MSM wants the shortest
// dependent chain (K = 8); inside prove() the reduction of one MSM hides under the accumulation of others, so fewer,
// longer segments (fewer small scalar multiplications) cost less MAD issue overall.
void msm_plan_set_segment(MsmPlan& p, int K);

constexpr int MSM_MAX_WINDOWS = 64;
// sorted entry = term index | window << 26 | sign << 31 over window tables (index < 2^26, window < 32), term index | sign << 31
// otherwise: msm_enqueue_batch refuses anything else with SONIC_ERR_INVALID_ARG (srs_msm_plan never plans it)
constexpr long MSM_TABLE_MAX_TERMS = 1L << 26;
constexpr int MSM_TABLE_MAX_WINDOWS = 32;
constexpr long MSM_MAX_TERMS = 1L << 31;

// Per-MSM hand-off between the bulk kernels and the (deferred, batched) tail: the per-window sums.
// Two forms: W window sums of a c-bit Horner walk (per-window bucket sets; pad1 == 0), or -- shared bucket sets reduced by the
// bit-sum butterfly, pad1 == 1 -- win[j] = sum of the buckets whose index has bit j set (j < W), win[W] = the sum of all buckets
// and pad0 = its weight: the MSM is sum_j 2^j win[j] + pad0 win[W].  pad1 == 2: an endomorphism pair -- the same form twice, the
// second at win[32 ..], result = first + phi(second).  msm_finish_host folds all of them.
struct MsmSlot {
  int W, c, pad0, pad1;
  G1XYZZ win[MSM_MAX_WINDOWS];
};

struct MsmWorkspace {
  DevBuf count, off, digits, entries, buckets, segres, scan_tmp, order, heavy_meta, heavy_partial, endo_scalars;
  // n = total terms over the k jobs of a batch
  void reserve(long n, const MsmPlan& pl, int k = 1);
};

// Queues one MSM on `st`: sum_i scalars[i] * points[i].  `scalars_mont` tells whether the Fr
// values are in Montgomery form (polynomial coefficient arrays are) or standard form (ABI inputs).
// Writes the W per-window sums into *d_slot.
void msm_enqueue(hipStream_t st, MsmWorkspace& ws, const MsmPlan& pl, PointArray d_points,
                 const Fr* d_scalars, long n, bool scalars_mont, MsmSlot* d_slot);

// Several MSMs as ONE kernel chain (shared-bucket plans over window tables only): job j owns bucket set j, so the
// latency-bound phases (running sums, trees, sort passes) run k times wider instead of k times in a row.  The jobs may
// differ in size, points and scalars; they share the plan (c, W, table stride).  k <= MSM_MAX_JOBS.
constexpr int MSM_MAX_JOBS = 16;     // (16 since round 6: the 7 + 4Q MSMs of a Q = 2 proof as ONE chain, prove.hip `fused`)
constexpr int MSM_ENDO_SLOT_OFFSET = 32;      // where the second half of an endomorphism pair sits in MsmSlot::win
// (the jobs of a batch share the byte stride between points; table_stride != 0: this job's window tables are that many points apart instead
// of the plan's -- the SRS's symmetric-sum tables hold d + 1 points per window where the bases hold 2d + 1)
struct MsmJob { PointArray points; const Fr* scalars; long n; MsmSlot* slot; long table_stride = 0; };
bool msm_can_batch(const MsmPlan& pl);
// ext_buckets (k == 1, shared-bucket plan): the chain stops after the accumulation and leaves the NB bucket sums there
// instead of reducing them -- the first half of an MSM whose buckets are sharded across ranks.
void msm_enqueue_batch(hipStream_t st, MsmWorkspace& ws, const MsmPlan& pl, const MsmJob* jobs, int k, bool scalars_mont,
                       G1XYZZ* ext_buckets = nullptr);
// The second half on the rank that owns bucket range [base, base + len): adds the k slices it received element-wise and
// reduces them with the range's weights into d_slot (one window sum).  len and base are multiples of MSM_SLICE_QUANTUM.
constexpr long MSM_SLICE_QUANTUM = 16384;     // = 2048 segments: whole 256-segment groups and wave-uniform segment bits
void msm_reduce_slices_enqueue(hipStream_t st, MsmWorkspace& ws, const G1XYZZ* d_slices, int k, long len, long base, int c, MsmSlot* d_slot);

// Host tail: Horner over the slot's window sums -> un-normalised XYZZ sum.
G1XYZZ msm_finish_host(const MsmSlot& s);
// canonical 96-byte encodings on the host: g1_host.hpp
int msm_window_override();
void msm_set_window_override(int c);

}  // namespace sonic
